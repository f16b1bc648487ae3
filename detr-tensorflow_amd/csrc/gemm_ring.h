// gemm_ring.h -- the "ring" form of the bf16 tile GEMM (round 5): 8 waves per workgroup, operands straight from global memory
// into LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`) through a ring of 64-deep K stages, counted `s_waitcnt vmcnt(N)` + raw
// `s_barrier`, and a ROW-BALANCED tile grid.  Same arithmetic as gemm_bf16c_body (v_mfma_f32_32x32x16_bf16, every accumulator
// sees its k-steps in ascending order): results are bit-identical to the 4-wave engine; same epilogue (gemm_core.h).
//
// Why.  The 4-wave engine pulls its tiles out of L2 at 13-16 B/clk/CU: 128x128 tiles read A twice
// and B 263 times (M = 33600), two workgroups per CU alternate short dependent phases (request -> wait -> ds_write -> barrier ->
// fragments -> MFMA), the 526 tiles of DETR's "just past a power of two" row counts cost a second partial round (+21 %), and a
// workgroup's epilogue overlaps nothing of its own.  Here:
//   * a workgroup owns `tile_rows` x BN outputs with BN = N (<= 256: row-complete, A is read from HBM exactly once) or 256-column
//     panels; `tile_rows` is NOT the tile capacity: the host picks the row pitch so that the launch is whole rounds of 256
//     workgroups (M = 33600: 255 workgroups of 132 rows instead of 263 + 263 of 128), and a wave skips the 32-row blocks its
//     pitch does not reach (wave-uniform), so the padding costs no MFMA issue;
//   * 8 waves as 2 (M) x 4 (N); a wave holds TM x TN 32x32 accumulator blocks (TM <= 4, TN <= 2: 128 x 64 per wave at most);
//   * no VGPR staging and no ds_write pass: a stage is filled by 1 KB DMA pieces (8 rows x 128 B of a K-contiguous operand, 8
//     sub-blocks of a transpose-read image), TM + 2 TN pieces per wave and stage, NS - 1 stages in flight across the barriers;
//   * the DMA writes lane-linearly, so the LDS image is chosen by the SOURCE address each lane requests (guide rule 21):
//     K-contiguous rows are 128 B = eight 16-byte chunks, chunk c of row r sits at slot 8 r + (c ^ ((r >> 1) & 7)); the 16 lanes of
//     every ds_read_b128 service group (rows distinct mod 16, one chunk index) then hit 16 distinct 16-byte slots of the 256-byte
//     bank row: conflict-free without padding (which a lane-linear DMA could not produce).  [k][n] operands keep the
//     transpose-read image of gemm_bf16_core.h ([4 k][16 n] sub-blocks, ds_read_b64_tr_b16).
// Pipeline of one tile (every wave issues its share of both operands' pieces; requests are unconditional -- past the last K
// stage the descriptor is empty: no traffic, zeros into a stage nobody reads -- so the counted waits are exact on every path):
//     prologue   issue stages 0 .. NS-2
//     step t     s_waitcnt vmcnt((NS-2) * pieces)   -> this wave's pieces of stage t have landed
//                s_barrier (raw: LDS-DMA requests stay in flight across it)  -> everybody's have, everybody is done with t-1
//                issue stage t+NS-1 into the ring slot stage t-1 occupied
//                fragments of stage t, MFMAs; sched_barrier so that no fragment read or MFMA drifts past the next barrier
// Eligibility (host, gemm_f32.hip): compute = bf16, A and B bf16 in memory, A K-contiguous, no batch / split-K / row sums,
// K % 64 == 0, 16-byte aligned rows.
#pragma once
#include "gemm_kernels.h"

namespace detr {

typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int RING_BK = 64;
constexpr int RING_THREADS = 512;
constexpr int RING_STAGE_ROW = RING_BK * 2;          // bytes per K-contiguous stage row

struct RingArgs {
    GemmArgs g;
    int tile_rows;       // row pitch of the tile grid (<= 64 * TM)
    int a_rows8;         // rows of the A stage image: tile_rows rounded up to a DMA piece (8 rows)
    int stage_bytes;     // a_rows8 * 128 + BN * 128
    int dump_off;        // byte offset of the 1 KB dump area behind the ring (pieces past the A image land there)
};

// host side (gemm_ring.hip)
struct RingPlan {
    int tm, tn, ns;                  // 32-row blocks per wave (tile capacity 64 tm rows), 32-column blocks per wave (BN = 128 tn), ring stages
    int tile_rows, tiles_m, tiles_n, wgs;
    int a_rows8, stage_bytes, dump_off, lds_bytes;
    double cost;
};
bool gemm_ring_plan(int M, int N, int K, RingPlan &p);
int gemm_ring_launch(const GemmArgs &g, bool b_kcontig, const RingPlan &p, hipStream_t s);

template <int N>
__device__ __forceinline__ void ring_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// K-contiguous operand [rows][k]: piece P = rows 8P .. 8P+7 of the stage image, lane (r = lane / 8, slot position lane % 8)
template <int NP>
struct RingDmaK {
    const unsigned short *base;
    long long text;            // bytes from `base` to the end of the operand
    unsigned voff[NP];         // loop-invariant per-lane offsets (BUF_OOB: row outside)
    int lds_off[NP];           // wave-uniform byte offset of piece i inside the stage image, < 0: dump area
    // rows [row0, row_end) of the operand belong to this tile; img_rows8: rows the stage image holds
    __device__ __forceinline__ void init(const void *p, long long ld, int row0, int row_end, int rows_total, int K, int img_rows8,
                                         int lane, int wave) {
        base = reinterpret_cast<const unsigned short *>(p);
        text = ((long long)(rows_total - 1) * ld + K) * 2;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int P = wave + 8 * i;
            const int r = 8 * P + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);                     // source chunk that belongs at this lane's slot
            const int g = row0 + r;
            voff[i] = (g < row_end) ? (unsigned)((long long)g * ld * 2) + 16u * (unsigned)c : BUF_OOB;
            lds_off[i] = (8 * P < img_rows8) ? P * 1024 : -1;
        }
    }
    __device__ __forceinline__ void issue(int k0, int K, char *img, char *dump) const {
        long long left = (k0 < K) ? text - 2ll * k0 : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(base + k0), 0, (int)(unsigned)left, 0x00020000);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            char *dst = lds_off[i] >= 0 ? img + lds_off[i] : dump;         // wave-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)dst, 16, voff[i], 0, 0, 0);
        }
    }
};

// MN-contiguous operand [k][mn], transpose-read image: sub-block (kb, ib) = 4 k x 16 mn at ((kb * NB + ib) * 128) bytes; piece P =
// sub-blocks 8P .. 8P+7; lane: sub-block 8P + lane / 8, k row (lane >> 1) & 3, 8-column half lane & 1
template <int BMN>
struct RingDmaMN {
    static constexpr int NB = BMN / 16;
    static constexpr int NP = BMN / 64;
    const unsigned short *base;
    long long text;
    unsigned ld2b;
    unsigned voff[NP];
    int wave;
    __device__ __forceinline__ void init(const void *p, long long ld, int mn0, int MN, int K, int lane, int wave_) {
        wave = wave_;
        base = reinterpret_cast<const unsigned short *>(p);
        text = ((long long)(K - 1) * ld + MN) * 2;
        ld2b = (unsigned)(ld * 2);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int sb = 8 * (wave + 8 * i) + (lane >> 3);
            const int kb = sb / NB, ib = sb % NB;
            const int k = 4 * kb + ((lane >> 1) & 3);
            const int col = mn0 + 16 * ib + 8 * (lane & 1);
            voff[i] = (col + 8 <= MN) ? (unsigned)k * ld2b + 2u * (unsigned)col : BUF_OOB;
        }
    }
    __device__ __forceinline__ void issue(int k0, int K, char *img, char *) const {
        long long left = (k0 < K) ? text - (long long)k0 * ld2b : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(base) + (long long)k0 * (ld2b >> 1), 0,
                                                                           (int)(unsigned)left, 0x00020000);
#pragma unroll
        for (int i = 0; i < NP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(img + (wave + 8 * i) * 1024), 16, voff[i], 0, 0, 0);
    }
};

// MFMA fragment (8 consecutive k of row row_base + (lane & 31)) out of a transpose-read image with NB sub-blocks per k group
template <int NB>
__device__ __forceinline__ bf16x8 ring_frag_tr(const char *img, int row_base, int ks, int lane) {
    const int g = lane >> 4, t = lane & 15;
    const int ib = (row_base >> 4) + (g & 1);
    const int kb = (ks >> 2) + 2 * (g >> 1);
    const unsigned short *p = reinterpret_cast<const unsigned short *>(img) + ((kb * NB + ib) * 64 + t * 4);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + NB * 64));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// one 64-deep stage: NM (<= TM) row blocks of this wave x TN column blocks x 4 k-steps
template <int NM, int TM, int TN, bool BKC>
__device__ __forceinline__ void ring_mma_stage(const char *As, const char *Bs, f32x16 (&acc)[TM][TN], int a_lane, int b_lane, int wn,
                                               const int (&xo)[4], int lane) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        bf16x8 a[NM], b[TN];
#pragma unroll
        for (int mi = 0; mi < NM; ++mi) a[mi] = *reinterpret_cast<const bf16x8 *>(As + a_lane + mi * (32 * RING_STAGE_ROW) + xo[kk]);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            if constexpr (BKC) b[ni] = *reinterpret_cast<const bf16x8 *>(Bs + b_lane + ni * (32 * RING_STAGE_ROW) + xo[kk]);
            else b[ni] = ring_frag_tr<TN * 8>(Bs, wn * (32 * TN) + ni * 32, kk * 16, lane);
        }
#pragma unroll
        for (int mi = 0; mi < NM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
}
template <int TM, int TN, bool BKC, int NS>
__device__ __forceinline__ void gemm_ring_body(const RingArgs &ra, const int id) {
    constexpr int BM = 64 * TM, BN = 128 * TN, WGM = 2, WGN = 4;
    using T = TileCfg<BM, BN, WGM, WGN>;
    static_assert(NS >= 2 && NS <= 4, "ring of 2 .. 4 stages");
    extern __shared__ __attribute__((aligned(1024))) char ring_smem[];
    const GemmArgs &g = ra.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: the DMA destinations and the block counts are wave-uniform)
    const int wm = wave / WGN, wn = wave % WGN;
    const int tn = id % g.tiles_n, tm = id / g.tiles_n;
    const int m0 = tm * ra.tile_rows, n0 = tn * BN;
    const int row_end = min(g.M, m0 + ra.tile_rows);                 // this tile's rows: [m0, row_end)
    // 32-row blocks this wave really has (wave-uniform): its rows start at m0 + wm * 32 TM
    const int my_rows = row_end - (m0 + wm * T::WTM);
    const int nmi = my_rows <= 0 ? 0 : (my_rows >= T::WTM ? TM : (my_rows + 31) >> 5);

    constexpr int NPA = TM, NPB = 2 * TN, PW = NPA + NPB;            // DMA pieces per wave and stage
    RingDmaK<NPA> la;
    la.init(g.A, g.lda, m0, row_end, g.M, g.K, ra.a_rows8, lane, wave);
    using LB = typename std::conditional<BKC, RingDmaK<NPB>, RingDmaMN<BN>>::type;
    LB lb;
    if constexpr (BKC) lb.init(g.B, g.ldb, n0, g.N, g.N, g.K, BN, lane, wave);
    else lb.init(g.B, g.ldb, n0, g.N, g.K, lane, wave);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int a_bytes = ra.a_rows8 * RING_STAGE_ROW;
    char *const dump = ring_smem + ra.dump_off;
    auto stage_a = [&](int s) { return ring_smem + s * ra.stage_bytes; };
    auto stage_b = [&](int s) { return ring_smem + s * ra.stage_bytes + a_bytes; };
    // fragment addressing: row (base multiple of 32) + (lane & 31) -> its swizzle term is ((lane & 31) >> 1) & 7 whatever the block
    const int l31 = lane & 31, h = lane >> 5;
    const int sw = (l31 >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xo[kk] = 16 * ((2 * kk + h) ^ sw);
    const int a_lane = (wm * T::WTM + l31) * RING_STAGE_ROW;
    const int b_lane = (wn * T::WTN + l31) * RING_STAGE_ROW;

    const int nkt = g.K / RING_BK;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
        la.issue(t * RING_BK, g.K, stage_a(t), dump);
        lb.issue(t * RING_BK, g.K, stage_b(t), dump);
    }
    // The K loop exists once per block count NM a wave can have (chosen ONCE, outside the loop: a per-stage switch made the
    // compiler shuffle all accumulators between the branches' register assignments every iteration)
    auto kloop = [&](auto NMC) {
        constexpr int NM = decltype(NMC)::value;
        int cur = 0, nxt = NS - 1;                      // ring slots of stage t and of stage t + NS - 1
        for (int t = 0; t < nkt; ++t) {
            ring_wait_vmcnt<(NS - 2) * PW>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            la.issue((t + NS - 1) * RING_BK, g.K, stage_a(nxt), dump);
            lb.issue((t + NS - 1) * RING_BK, g.K, stage_b(nxt), dump);
            if constexpr (NM > 0) ring_mma_stage<NM, TM, TN, BKC>(stage_a(cur), stage_b(cur), acc, a_lane, b_lane, wn, xo, lane);
            __builtin_amdgcn_sched_barrier(0);
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
    };
    if (nmi == TM) kloop(std::integral_constant<int, TM>{});
    else if (TM >= 2 && nmi == TM - 1) kloop(std::integral_constant<int, (TM >= 2 ? TM - 1 : 0)>{});
    else if (TM >= 3 && nmi == TM - 2) kloop(std::integral_constant<int, (TM >= 3 ? TM - 2 : 0)>{});
    else if (TM >= 4 && nmi == TM - 3) kloop(std::integral_constant<int, (TM >= 4 ? TM - 3 : 0)>{});
    else kloop(std::integral_constant<int, 0>{});
    ring_wait_vmcnt<0>();                               // the trailing empty requests still write (zeros) into the ring:
    __syncthreads();                                    // nobody reuses the array (epilogue staging) before they have landed
    epilogue<BM, BN, WGM, WGN>(acc, reinterpret_cast<float *>(ring_smem), g.C, g.ldc, row_end, g.N, m0, n0, wm, wn, lane, wave, g.e);
}

template <int TM, int TN, bool BKC, int NS>
__global__ __launch_bounds__(RING_THREADS) void gemm_ring_kernel(RingArgs ra) {
    gemm_ring_body<TM, TN, BKC, NS>(ra, xcd_remap((int)blockIdx.x, (int)gridDim.x));
}

}  // namespace detr
