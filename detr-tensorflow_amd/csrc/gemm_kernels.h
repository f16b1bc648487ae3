// gemm_kernels.h -- device side of the tile GEMM engine: kernel arguments, the fp32 and bf16-compute K loops and the kernel
// templates (plain, 64-deep-K and grouped launches).  Host-side planning / dispatch: gemm_f32.hip.  Split out so that a single
// instantiation can be compiled on its own for ISA inspection (scripts/isa/).
#pragma once
#include "gemm_core.h"
#include "gemm_bf16_core.h"

namespace detr {

struct GemmArgs {
    int M, N, K;
    const float *A; long long lda;
    const float *B; long long ldb;
    float *C; long long ldc;
    int batch_inner;
    long long sA0, sA1, sB0, sB1, sC0, sC1;
    int split_k;
    long long part_stride;   // partial-slab stride (deterministic split-K), 0 otherwise
    int tiles_m, tiles_n;
    int a_vec, b_vec;
    EpiArgs e;
    // optional fused bias gradient: row sums of the (MN-contiguous) A operand, i.e. rowsum[m] = sum_k A[m][k]
    float *rowsum;               // split_k == 1: rowsum[m] += rowsum_alpha * sum;  else partial slab [split][M]
    float rowsum_alpha;
    int rowsum_partial;
    int b16;                     // B operand is bf16 in memory (bf16 compute only)
    int a16;                     // A operand is bf16 in memory (bf16 activation storage)
    int split_xcd;               // split-K: remap the whole (split, tile) space over the XCDs (0 = per-split tile remap only, A/B hook)
    int slab_ts;                 // split-K partials as tile-ordered slabs (gemm_core.h: store_slab_ts); part_stride counts padded tiles
    LnArgs ln;                   // gemm_bf16c_ln_kernel only: the LayerNorm fused behind the row-complete tile (gemm_core.h: epilogue_ln)
};

// Row sums of A collected from the loader registers (fp32, before any rounding): every thread owns the float4 of
// one column group (4 consecutive rows m of A); the holders of a group are combined in a fixed order through LDS.
// bf16_map: 0 = LoaderMN (fp32, K tile 16), 1 = LoaderMNt / narrow LoaderMNth unit map, 2 = WIDE LoaderMNth (two slots per
// thread: registers 2i -> column group 4*ib + c, 2i + 1 -> 4*ib + c + 1)
template <int BM, int NSLOT>
__device__ __forceinline__ void rowsum_finish(const float4 (&rs)[NSLOT], float *lds, const GemmArgs &g, int m0, int split,
                                              int tid, int bf16_map) {
    float4 *part = reinterpret_cast<float4 *>(lds);            // [NSLOT][256]
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) part[i * 256 + tid] = rs[i];
    __syncthreads();
    if (tid < BM) {
        const int cg = tid >> 2, comp = tid & 3;
        float sum = 0.0f;
        if (bf16_map == 2) {     // wide LoaderMNth: group 4*ib + c' lives in slot c' & 1 of threads (c' >> 1) + 2*kr + 8*ib + 8*NB*h
            constexpr int NB = BM / 16;
            const int ib = cg >> 2, cp = cg & 3;
            const float *src = lds + ((cp & 1) * 256 + (cp >> 1) + 8 * ib) * 4 + comp;
#pragma unroll
            for (int h = 0; h < 32 / NB; ++h)
#pragma unroll
                for (int kr = 0; kr < 4; ++kr) sum += src[(2 * kr + 8 * NB * h) * 4];
        } else if (bf16_map == 1) {          // LoaderMNt: group 4*ib + c lives in threads c + 4*kr + 16*ib + 16*NB*h (kr < 4, h < 16/NB)
            constexpr int NB = BM / 16;
            const float *src = lds + ((cg & 3) + 16 * (cg >> 2)) * 4 + comp;
#pragma unroll
            for (int h = 0; h < 16 / NB; ++h)
#pragma unroll
                for (int kr = 0; kr < 4; ++kr) sum += src[(4 * kr + 16 * NB * h) * 4];
        } else {                 // LoaderMN: group t % (BM/4) lives in threads cg + (BM/4) * j
            constexpr int VPR = BM / 4;
            const float *src = lds + cg * 4 + comp;
#pragma unroll
            for (int j = 0; j < 256 / VPR; ++j) sum += src[j * VPR * 4];
        }
        const int m = m0 + tid;
        if (m < g.M) {
            if (g.rowsum_partial) g.rowsum[(long long)split * g.M + m] = sum;
            else g.rowsum[m] += g.rowsum_alpha * sum;
        }
    }
    __syncthreads();
}

constexpr int GEMM_MAX_GROUP = 4;
struct GemmGroupArgs {       // up to 4 independent GEMMs of one kernel variant in ONE launch
    GemmArgs g[GEMM_MAX_GROUP];
    // flattened work list: workgroup L (after the XCD remap) is item L - work_off[m] of member m, item = split * tiles + tile
    int work_off[GEMM_MAX_GROUP + 1];
    int n;
};

// Which (tile, z) a workgroup of a PLAIN launch computes.  No split-K (or batched): the tiles of each z are remapped so that an
// XCD walks a contiguous run (N fastest: an A row-panel is fetched into one L2).  Split-K weight gradients: the whole
// (split, tile) space is remapped as ONE list, so that the tiles of a split run on ONE XCD back to back -- that K slice of A
// and B then enters one L2 once instead of every L2 (measured over-fetch of the 1x1-conv / Linear weight gradients with the
// per-z remap: 2.1x - 3.0x the algorithmic bytes).
__device__ __forceinline__ void gemm_work_item(const GemmArgs &g, int &tile, int &z) {
    if (g.split_k > 1 && (int)gridDim.z == g.split_k && g.split_xcd) {
        const int id = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.z), (int)(gridDim.x * gridDim.z));
        tile = id % (int)gridDim.x;
        z = id / (int)gridDim.x;
    } else {
        tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
        z = (int)blockIdx.z;
    }
}
__device__ __forceinline__ bool gemm_group_item(const GemmGroupArgs &G, int &m, int &tile, int &z) {
    const int L = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    if (L >= G.work_off[G.n]) return false;
    m = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_GROUP; ++i)
        if (i < G.n && L >= G.work_off[i]) m = i;
    const int r = L - G.work_off[m];
    const int nwg = G.g[m].tiles_m * G.g[m].tiles_n;
    tile = r % nwg;
    z = r / nwg;
    return true;
}

template <int BM, int BN, int WGM, int WGN, bool AK, bool BKC, bool SPLIT3 = false>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs &g, const int id, const int zidx) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    __shared__ __attribute__((aligned(16))) char smem_raw[SmemBytes<BM, BN, WGN>::VALUE];
    GemmSmem<BM, BN> &sm = *reinterpret_cast<GemmSmem<BM, BN> *>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    const int tn = id % g.tiles_n, tm = id / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = zidx;
    const int split = z % g.split_k;
    const int zb = z / g.split_k;
    const int z0 = zb / g.batch_inner, z1 = zb % g.batch_inner;
    const float *A = g.A + z0 * g.sA0 + z1 * g.sA1;
    const float *B = g.B + z0 * g.sB0 + z1 * g.sB1;
    float *C = g.C + z0 * g.sC0 + z1 * g.sC1 + (long long)split * g.part_stride;

    const int nkt = (g.K + GEMM_BK - 1) / GEMM_BK;
    const int per = (nkt + g.split_k - 1) / g.split_k;
    const int kt0 = split * per;
    const int kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;   // empty split (uniform per workgroup, before any barrier)

    using LA = typename std::conditional<AK, LoaderK<BM>, LoaderMN<BM>>::type;
    using LB = typename std::conditional<BKC, LoaderK<BN>, LoaderMN<BN>>::type;
    LA la;
    LB lb;
    la.init(A, g.lda, m0, g.M, g.K, g.a_vec != 0, tid);
    lb.init(B, g.ldb, n0, g.N, g.K, g.b_vec != 0, tid);

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const bool do_rs = !AK && g.rowsum != nullptr && tn == 0;      // workgroup-uniform
    float4 rs[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
    auto rs_add = [&](const float4 (&r)[LA::NV]) {
#pragma unroll
        for (int i = 0; i < LA::NV; ++i) { rs[0].x += r[i].x; rs[0].y += r[i].y; rs[0].z += r[i].z; rs[0].w += r[i].w; }
    };
    float4 ra[LA::NV], rb[LB::NV];
    la.load(kt0 * GEMM_BK, g.K, ra);
    lb.load(kt0 * GEMM_BK, g.K, rb);
    if (do_rs) rs_add(ra);
    la.store(sm.A[0], ra);
    lb.store(sm.B[0], rb);
    __syncthreads();

    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1) < kt1;
        if (more) {
            la.load((kt + 1) * GEMM_BK, g.K, ra);
            lb.load((kt + 1) * GEMM_BK, g.K, rb);
        }
        mma_ktile_sel<BM, BN, WGM, WGN, SPLIT3>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
        if (more) {
            if (do_rs) rs_add(ra);
            la.store(sm.A[cur ^ 1], ra);
            lb.store(sm.B[cur ^ 1], rb);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (do_rs) rowsum_finish<BM, 1>(rs, reinterpret_cast<float *>(smem_raw), g, m0, split, tid, 0);
    if constexpr (WGM == 2 && WGN == 2) {
        if (g.slab_ts) {             // (kernel argument: uniform over the grid)
            store_slab_ts<BM, BN, WGM, WGN>(acc, C + (long long)id * (BM * BN), wave, lane);
            return;
        }
    }
    epilogue<BM, BN, WGM, WGN, false>(acc, reinterpret_cast<float *>(smem_raw), C, g.ldc, g.M, g.N, m0, n0, wm, wn, lane, wave, g.e);
}

// SPLIT3: the K tiles run on the bf16 matrix pipe at fp32 accuracy (gemm_core.h: mma_ktile_split3; detr_gemm_desc.compute = 2)
template <int BM, int BN, int WGM, int WGN, bool AK, bool BKC, bool SPLIT3 = false>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_kernel(GemmArgs g) {
    int tile, z;
    gemm_work_item(g, tile, z);
    gemm_f32_body<BM, BN, WGM, WGN, AK, BKC, SPLIT3>(g, tile, z);
}
// grouped launch: the members share the kernel variant; workgroups past a member's own tile / split count retire
template <int BM, int BN, int WGM, int WGN, bool AK, bool BKC, bool SPLIT3 = false>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_group_kernel(GemmGroupArgs G) {
    int m, tile, z;
    if (!gemm_group_item(G, m, tile, z)) return;
    gemm_f32_body<BM, BN, WGM, WGN, AK, BKC, SPLIT3>(G.g[m], tile, z);
}

// bf16-compute variant (fp32 storage): same arguments, same epilogue, operands rounded to bf16 into LDS.
// BK = 64 (all-bf16 operands only): the same pipeline over K tiles twice as deep -- half the barrier-separated iterations,
// each with twice the MFMA work behind one LDS round trip.  The split ranges stay in units of 32 (the host's slab arithmetic,
// gemm_effective_split, does not depend on the variant); a range that is not a multiple of 64 ends in a half-empty tile whose
// missing half is never requested (out-of-range offsets -> zeros).
template <int BM, int BN, int WGM, int WGN, bool AK, bool BKC, bool A16, bool B16, int BK = BF_BK, bool LN = false>
__device__ __forceinline__ void gemm_bf16c_body(const GemmArgs &g, const int id, const int zidx) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    static_assert(BK == BF_BK || (A16 && B16), "the 64-deep K tile exists for bf16 operands only");
    __shared__ __attribute__((aligned(16))) char smem_raw[BfSmemBytes<BM, BN, WGN, BK>::VALUE];
    BfSmem<BM, BN, BK> &sm = *reinterpret_cast<BfSmem<BM, BN, BK> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tn = id % g.tiles_n, tm = id / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = zidx;
    const int split = z % g.split_k;
    const int zb = z / g.split_k;
    const int z0 = zb / g.batch_inner, z1 = zb % g.batch_inner;
    const float *A = g.A + z0 * g.sA0 + z1 * g.sA1;
    const float *B = g.B + z0 * g.sB0 + z1 * g.sB1;
    float *C = g.C + z0 * g.sC0 + z1 * g.sC1 + (long long)split * g.part_stride;
    const int nkt = (g.K + BF_BK - 1) / BF_BK;
    const int per = (nkt + g.split_k - 1) / g.split_k;
    const int kt0_32 = split * per;
    const int kt1_32 = min(nkt, kt0_32 + per);
    if (kt0_32 >= kt1_32) return;
    const int kbeg = kt0_32 * BF_BK;
    const int kend = min(g.K, kt1_32 * BF_BK);   // requests past this split's K range resolve to the out-of-range offset: no traffic
    const int kt0 = 0, kt1 = (kend - kbeg + BK - 1) / BK;      // K tiles of THIS variant's depth, relative to kbeg

    // MN-contiguous: transpose-read image.  A16: the A operand is bf16 in memory (bf16 activation storage)
    using LA = typename std::conditional<A16, typename std::conditional<AK, LoaderKh<BM, BK>, LoaderMNth<BM, true, BK>>::type,
                                         typename std::conditional<AK, LoaderKb<BM>, LoaderMNt<BM>>::type>::type;
    // B16: the B operand is already bf16 in memory (per-step weight shadow): half the bytes, no conversion
    using LB = typename std::conditional<B16, typename std::conditional<BKC, LoaderKh<BN, BK>, LoaderMNth<BN, true, BK>>::type,
                                         typename std::conditional<BKC, LoaderKb<BN>, LoaderMNt<BN>>::type>::type;
    constexpr int NRA = LA::NREG;
    constexpr int NRB = LB::NREG;
    LA la;
    LB lb;
    la.init(A, g.lda, m0, g.M, g.K, g.a_vec != 0, tid);
    lb.init(B, g.ldb, n0, g.N, g.K, g.b_vec != 0, tid);
    if constexpr (DETR_KLOOP_PIPE != 0 && DETR_ABLATE == 0) {       // per-tile descriptors of the bf16-storage loaders
        if constexpr (A16 && AK) la.init_tiles(A, g.lda, g.M, g.K);
        if constexpr (A16 && !AK) la.init_tiles(A, g.lda, kend);
        if constexpr (B16 && BKC) lb.init_tiles(B, g.ldb, g.N, g.K);
        if constexpr (B16 && !BKC) lb.init_tiles(B, g.ldb, kend);
    }
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const bool do_rs = !AK && g.rowsum != nullptr && tn == 0;      // workgroup-uniform
    constexpr int NRS = (!AK && A16) ? 2 : 1;      // wide bf16 loader: a thread's even / odd registers belong to two column groups
    float4 rs[NRS];
#pragma unroll
    for (int i = 0; i < NRS; ++i) rs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto rs_add = [&](const typename LA::Reg (&r)[NRA]) {     // fp32: every register of a thread belongs to the column group 4*ib + c of its tid
        if constexpr (!AK && !A16) {
#pragma unroll
            for (int i = 0; i < NRA; ++i) { rs[0].x += r[i].x; rs[0].y += r[i].y; rs[0].z += r[i].z; rs[0].w += r[i].w; }
        } else if constexpr (!AK && A16) {      // bf16 storage (LoaderMNth: the same unit map as LoaderMNt, 4 bf16 = 4 consecutive rows m)
#pragma unroll
            for (int i = 0; i < NRA; ++i) {
                float4 &t = rs[i & 1];
                t.x += bf16_bits_to_f32(r[i].x & 0xFFFFu); t.y += __builtin_bit_cast(float, r[i].x & 0xFFFF0000u);
                t.z += bf16_bits_to_f32(r[i].y & 0xFFFFu); t.w += __builtin_bit_cast(float, r[i].y & 0xFFFF0000u);
            }
        }
    };
    // Operand pipeline, two K tiles deep: while tile kt is multiplied out of LDS, tile kt+1 sits in one register set
    // (requested an iteration ago, stored to the other LDS buffer at the top of this iteration) and tile kt+2 is in
    // flight into the second set.  The barrier orders LDS only (lds_barrier, common.h): __syncthreads() would drain
    // vmcnt and with it the requests that are supposed to stay in flight -- with it, the iteration time of a workgroup
    // was one HBM / L2 round trip however deep the register pipeline.
    // (Round 4 measured FOUR register sets on the 64x64 tiles -- twice the bytes in flight: M800 N256 K2048 23 -> 19 us, every short-K
    //  shape 5-8 % slower (longer prologue), M8400 N256 K2048 unchanged, step +0.08 ms: not kept.  The launches are not bound by
    //  the request depth either; see NOTEBOOK 7c.)
    constexpr int DEPTH = 2;
    typename LA::Reg ra0[NRA], ra1[NRA];
    typename LB::Reg rb0[NRB], rb1[NRB];
    la.load(kbeg + kt0 * BK, kend, ra0);
    lb.load(kbeg + kt0 * BK, kend, rb0);
    if (do_rs) rs_add(ra0);
    la.store(sm.A[0], ra0);
    lb.store(sm.B[0], rb0);
    // (the requests are unconditional -- past the last tile they fall outside the buffer descriptor or fetch a tile
    //  that is never stored: with conditional requests the compiler cannot tell how many are outstanding and waits for
    //  vmcnt(0) before every LDS store, which puts the full round trip back into each iteration)
    la.load(kbeg + (kt0 + 1) * BK, kend, ra0);
    lb.load(kbeg + (kt0 + 1) * BK, kend, rb0);
    la.load(kbeg + (kt0 + 2) * BK, kend, ra1);
    lb.load(kbeg + (kt0 + 2) * BK, kend, rb1);
    lds_barrier();
    // one iteration: `rp` holds tile kt+1 (stored now, then refilled with tile kt+1+DEPTH), LDS[cur] holds tile kt
    auto iter = [&](const int kt, const int cur, typename LA::Reg (&rpa)[NRA], typename LB::Reg (&rpb)[NRB]) {
        if (do_rs && kt + 1 < kt1) rs_add(rpa);
        // (128x128 tiles with an fp32-storage operand keep the compiler's order: their 168-register budget has no room for the
        //  second fragment set -- the pipelined form spilled up to 360 bytes there; no launch of the step uses them)
        if constexpr (DETR_KLOOP_PIPE != 0 && DETR_ABLATE == 0 && ((A16 && B16) || BM * BN < 128 * 128)) {
            // explicitly pipelined iteration (gemm_bf16_core.h: KPipe): same stores, requests and MFMA order per accumulator
            using P = KPipe<BM, BN, WGM, WGN, !AK, !BKC, BK>;
            typename P::Frags f0, f1;
            P::read(sm.A[cur], sm.B[cur], 0, wm, wn, lane, f0);
            P::read(sm.A[cur], sm.B[cur], 16, wm, wn, lane, f1);
            P::mma(f0, acc);
            la.store(sm.A[cur ^ 1], rpa);
            lb.store(sm.B[cur ^ 1], rpb);
            if constexpr (P::NS > 2) P::read(sm.A[cur], sm.B[cur], 32, wm, wn, lane, f0);
            P::mma(f1, acc);
            if constexpr (A16) la.load_tile(kbeg + (kt + 1 + DEPTH) * BK, kend, rpa);
            else la.load(kbeg + (kt + 1 + DEPTH) * BK, kend, rpa);
            if constexpr (B16) lb.load_tile(kbeg + (kt + 1 + DEPTH) * BK, kend, rpb);
            else lb.load(kbeg + (kt + 1 + DEPTH) * BK, kend, rpb);
            // steps 2 .. NS-1: fragments of step s+1 are read before the MFMAs of step s (two sets, alternating)
#pragma unroll
            for (int st = 2; st < P::NS; ++st) {
                if (st + 1 < P::NS) {
                    if (st & 1) P::read(sm.A[cur], sm.B[cur], 16 * (st + 1), wm, wn, lane, f0);
                    else P::read(sm.A[cur], sm.B[cur], 16 * (st + 1), wm, wn, lane, f1);
                }
                if (st & 1) P::mma(f1, acc);
                else P::mma(f0, acc);
            }
            constexpr int NW = LA::NDSW + LB::NDSW, NG = LA::NVMEM + LB::NVMEM;
            sgb_ds_read<2 * P::READS>();
            sgb_mfma_block<P::MF, NW, 0, 0>();                 // step 0 + the next tile's LDS stores
            if constexpr (P::NS > 2) sgb_ds_read<P::READS>();
            sgb_mfma_block<P::MF, 0, NG, 3 * NG>();            // step 1 + the requests of the tile after next (3 VALU per offset)
            sgb_tail<P::NS - 2, P::READS, P::MF>();
            lds_barrier();
            return;
        }
        if constexpr ((DETR_ABLATE & 4) == 0) {
            la.store(sm.A[cur ^ 1], rpa);          // (unconditional as well: after the last tile it writes a buffer nobody reads)
            lb.store(sm.B[cur ^ 1], rpb);
        } else {
            for (int i = 0; i < NRA; ++i) ablate_keep(rpa[i]);
            for (int i = 0; i < NRB; ++i) ablate_keep(rpb[i]);
        }
        if constexpr ((DETR_ABLATE & 2) == 0) {
            la.load(kbeg + (kt + 1 + DEPTH) * BK, kend, rpa);
            lb.load(kbeg + (kt + 1 + DEPTH) * BK, kend, rpb);
        }
        mma_ktile_bf16<BM, BN, WGM, WGN, !AK, !BKC, BK>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
#ifdef DETR_IGLP
        __builtin_amdgcn_iglp_opt(DETR_IGLP);
#endif
        if constexpr ((DETR_ABLATE & 8) == 0) lds_barrier();
    };
    {   // whole pairs in the loop, an odd last tile after it: every path into the loop header carries the same
        // sequence of outstanding requests, so the compiler's vmcnt waits are exact (vmcnt(2) / vmcnt(3) before the stores)
        int kt = kt0;
        for (; kt + 2 <= kt1; kt += 2) {
            iter(kt, 0, ra0, rb0);
            iter(kt + 1, 1, ra1, rb1);
        }
        if (kt < kt1) iter(kt, 0, ra0, rb0);
    }
    __syncthreads();
    if (do_rs) rowsum_finish<BM, NRS>(rs, reinterpret_cast<float *>(smem_raw), g, m0, split, tid, (!AK && A16) ? 2 : 1);
    if constexpr (LN) {              // row-complete 32 x 256 tile: C, then the LayerNorm of its rows (epilogue_ln)
        static_assert(BM == LN_TILE_M && BN == LN_TILE_N && WGM == 1 && WGN == 4, "the fused LayerNorm needs the row-complete tile");
        static_assert(LN_TILE_M * LN_STAGE_LD * 4 <= BfSmemBytes<BM, BN, WGN, BK>::VALUE, "staged tile must fit the operand LDS");
        epilogue_ln(acc, reinterpret_cast<float *>(smem_raw), C, g.ldc, g.M, m0, wn, lane, wave, g.e, g.ln);
        return;
    }
    if constexpr (WGM == 2 && WGN == 2) {
        if (g.slab_ts) {             // split-K partial: the accumulator registers as they are, 16-byte lane-linear stores
            store_slab_ts<BM, BN, WGM, WGN>(acc, C + (long long)id * (BM * BN), wave, lane);
            return;
        }
    }
    epilogue<BM, BN, WGM, WGN>(acc, reinterpret_cast<float *>(smem_raw), C, g.ldc, g.M, g.N, m0, n0, wm, wn, lane, wave, g.e);
}

template <int BM, int BN, int WGM, int WGN, bool AK, bool BKC, bool A16, bool B16>
__global__ __launch_bounds__(GEMM_THREADS, (BM * BN >= 128 * 128) ? 3 : (BM * BN == 64 * 64 ? DETR_GEMM64_MINW : 1)) void gemm_bf16c_kernel(GemmArgs g) {
    int tile, z;
    gemm_work_item(g, tile, z);
    gemm_bf16c_body<BM, BN, WGM, WGN, AK, BKC, A16, B16>(g, tile, z);
}
// all-bf16 operands, 64-deep K tiles (36 KB / 72 KB of LDS: 4 / 2 workgroups per CU)
template <int BM, int BN, int WGM, int WGN, bool AK, bool BKC>
__global__ __launch_bounds__(GEMM_THREADS, (BM * BN >= 128 * 128) ? 2 : 4) void gemm_bf16c_k64_kernel(GemmArgs g) {
    int tile, z;
    gemm_work_item(g, tile, z);
    gemm_bf16c_body<BM, BN, WGM, WGN, AK, BKC, true, true, 64>(g, tile, z);
}
// GEMM + LayerNorm of the transformer blocks (N = 256): row-complete 32 x 256 tiles, [n][k] bf16 weights, A fp32 or bf16
template <bool A16>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16c_ln_kernel(GemmArgs g) {
    int tile, z;
    gemm_work_item(g, tile, z);
    gemm_bf16c_body<LN_TILE_M, LN_TILE_N, 1, 4, true, true, A16, true, BF_BK, true>(g, tile, z);
}
// (Round 4 measured a 128-deep K tile for the 64x64-tile launches with K >= 1024 -- half the barrier-separated iterations, 70 KB of
//  LDS, 2 workgroups per CU; bit-identical: M8400 N256 K2048 34.7 -> 40.2 us, M8400 N512 K2048 42.3 -> 50.3 us.  Not kept: these
//  launches are bound by the bytes their tiles pull out of L2, not by the number of iterations -- NOTEBOOK 7c.)
template <int BM, int BN, int WGM, int WGN, bool AK, bool BKC, bool A16, bool B16>
__global__ __launch_bounds__(GEMM_THREADS, (BM * BN == 64 * 64) ? DETR_GEMM64_MINW : 1) void gemm_bf16c_group_kernel(GemmGroupArgs G) {
    int m, tile, z;
    if (!gemm_group_item(G, m, tile, z)) return;
    gemm_bf16c_body<BM, BN, WGM, WGN, AK, BKC, A16, B16>(G.g[m], tile, z);
}

}  // namespace detr
