// gemm_ring.h -- the "ring" form of the bf16 tile GEMM (round 5): 8 waves per workgroup, operands straight from global memory
// into LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`) through a ring of 64-deep K stages, counted `s_waitcnt vmcnt(N)` + raw
// `s_barrier`, and a ROW-BALANCED tile grid.  Same arithmetic as gemm_bf16c_body (v_mfma_f32_32x32x16_bf16, every accumulator
// sees its k-steps in ascending order): results are bit-identical to the 4-wave engine; same epilogue (gemm_core.h).
//
// Why (NOTEBOOK 4b / 7c).  The 4-wave engine pulls its tiles out of L2 at 13-16 B/clk/CU: 128x128 tiles read A twice
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
//     bank row: conflict-free without padding (which a lane-linear DMA could not produce).  [k][n] operands: a transpose-read
//     image in row-major pieces (RingDmaMN below; fragments by ds_read_b64_tr_b16).
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
    int trace_off;       // -DDETR_ABLATE=64 builds: byte offset of the 1 KB timeline area behind everything else (scripts/experiments/ring_trace.py)
    int ablate;          // timing experiments only (DETR_HIP_RING_ABLATE; results are WRONG with any bit set): 1 no fragment reads / MFMAs,
                         // 2 no A requests, 4 no B requests, 8 no epilogue, 16 no K loop at all
};

// host side (gemm_ring.hip)
struct RingPlan {
    int tm, tn, ns;                  // 32-row blocks per wave (tile capacity 64 tm rows), 32-column blocks per wave (BN = 128 tn), ring stages
    int tile_rows, tiles_m, tiles_n, wgs;
    int a_rows8, stage_bytes, dump_off, lds_bytes;
    double cost;
};
bool gemm_ring_plan(int M, int N, int K, RingPlan &p);
// split-K weight gradient with tile-ordered slabs on 128 x 128 tiles (tiles_m / tiles_n / split_k / part_stride / slab_ts set by the caller)
int gemm_ring_wgrad_launch(const GemmArgs &g, int bm, int bn, int split, hipStream_t s);
int gemm_ring_launch(const GemmArgs &g, bool b_kcontig, const RingPlan &p, hipStream_t s);
// fp32 form (exact-f32 parity mode): plan for MFMA-bound work (whole 32-row blocks, per-CU balance), launch
bool gemm_ring_f32_plan(int M, int N, int K, RingPlan &p);
int gemm_ring_f32_launch(const GemmArgs &g, bool b_kcontig, const RingPlan &p, hipStream_t s);

#if (DETR_ABLATE & 64) != 0
// timeline experiment: wave 0 of three workgroups stamps s_memtime in front of the counted wait of every K stage, in front of the barrier and
// behind it; [workgroup][40 stages x 3 + kernel start, loop start, loop end, kernel end, 100 MHz counter at kernel start / end]
constexpr int RING_TR_STEPS = 40, RING_TR_N = RING_TR_STEPS * 3 + 6;
extern __device__ long long ring_trace[3][RING_TR_N];
#define RING_STAMP(idx) do { if (tr_w) { const long long t_ = __builtin_readcyclecounter(); if (lane == 0) trl[(idx)] = t_; } } while (0)
#define RING_STAMP_STEP(s, k) do { if ((s) < RING_TR_STEPS) RING_STAMP((s) * 3 + (k)); } while (0)
#else
#define RING_STAMP(idx) do { } while (0)
#define RING_STAMP_STEP(s, k) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void ring_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer (`rs`, per-lane byte offset voff) to LDS bytes [lds_addr, lds_addr + 1024).
// Inline assembly on purpose: hipcc's waitcnt pass knows that a `buffer_load ... lds` it emitted itself is a pending LDS write,
// and in front of every ds_read_b64_tr_b16 (the [k][n] fragments) it therefore waits for vmcnt(0) -- measured: the [k][n] shapes ran
// 6-17 us slower than the [n][k] ones, the ring drained at every k-step.  Which requests a read depends on is decided HERE,
// by the counted waits of the pipeline (ring_wait_vmcnt in front of the barrier that publishes a stage); requests the compiler
// does not know about can only make ITS OWN vmcnt waits (epilogue loads) wait longer, never shorter (the counter is in-order).
// M0 carries the LDS address (what the compiler emits for the builtin: s_add_i32 m0, <addr>, 0); one wait state between the M0
// write and the request (gfx9 hazard table: S_MOV M0 -> buffer ... lds).
__device__ __forceinline__ void ring_dma_piece(u32x4 rs, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0");
}
// the same request with the non-temporal hint (a stream that is read once: experiment DETR_HIP_RING_NT)
__device__ __forceinline__ void ring_dma_piece_nt(u32x4 rs, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0");
}
__device__ __forceinline__ u32x4 ring_rsrc(const void *base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    u32x4 r;
    r[0] = (unsigned)b;
    r[1] = (unsigned)(b >> 32) & 0xFFFFu;          // stride 0: raw buffer
    r[2] = bytes;
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ unsigned ring_lds_addr(char *p) {      // LDS byte address of a pointer into the workgroup's shared array
    return (unsigned)(unsigned long long)(lds_void_t *)p;
}

// K-contiguous operand [rows][k]: piece P = rows 8P .. 8P+7 of the stage image, lane (r = lane / 8, slot position lane % 8)
// ES: bytes per element (2: bf16, 64-deep stages; 4: fp32, 32-deep stages -- a stage row is 128 bytes either way)
#ifndef DETR_RING_ISSUE_STEPS
#define DETR_RING_ISSUE_STEPS 4   // k-steps of a stage (of 4) that carry its DMA requests (A/B builds: 1, 2, 3)
#endif
#ifndef DETR_RING_NT
#define DETR_RING_NT 0          // 1: the A requests of gemm_ring_kernel carry the non-temporal hint (A/B builds, scripts/experiments)
#endif
template <int NP, int ES = 2, bool NT = false>
struct RingDmaK {
    static constexpr int NPIECES = NP;
    const char *base;
    long long text;            // bytes from `base` to the end of the operand
    unsigned voff[NP];         // loop-invariant per-lane offsets (BUF_OOB: row outside)
    int lds_off[NP];           // wave-uniform byte offset of piece i inside the stage image
    int in_img[NP];            // -1: the piece lies inside the stage image, 0: past it (its zeros go to the dump area)
    // rows [row0, row_end) of the operand belong to this tile; img_rows8: rows the stage image holds
    __device__ __forceinline__ void init(const void *p, long long ld, int row0, int row_end, int rows_total, int K, int img_rows8,
                                         int lane, int wave) {
        base = reinterpret_cast<const char *>(p);
        text = ((long long)(rows_total - 1) * ld + K) * ES;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int P = wave + 8 * i;
            const int r = 8 * P + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);                     // source chunk that belongs at this lane's slot
            const int g = row0 + r;
            voff[i] = (g < row_end) ? (unsigned)((long long)g * ld * ES) + 16u * (unsigned)c : BUF_OOB;
            lds_off[i] = P * 1024;
            in_img[i] = (8 * P < img_rows8) ? -1 : 0;
        }
    }
    // descriptor of K stage k0: base advanced by k0, an EMPTY range past the last stage (no traffic); scalar instructions only
    __device__ __forceinline__ u32x4 stage_rsrc(int k0, int K) const {
        const unsigned left = (k0 < K) ? (unsigned)(text - (long long)ES * k0) : 0u;
        return ring_rsrc(base + (long long)ES * k0, left);
    }
    // piece I (a compile-time constant after inlining).  lds0: LDS byte address of the workgroup's array; img / dump: byte offsets into
    // it.  Branch-free on purpose (mask arithmetic instead of ?: -- pointer- and flag-typed selects compiled to scalar branches,
    // which cut the stage into several basic blocks).
    __device__ __forceinline__ void issue_one(const int I, u32x4 rs, unsigned lds0, int img, int dump) const {
        const int off = dump + (in_img[I] & (img + lds_off[I] - dump));      // wave-uniform
        if constexpr (NT) ring_dma_piece_nt(rs, lds0 + (unsigned)off, voff[I]);
        else ring_dma_piece(rs, lds0 + (unsigned)off, voff[I]);
    }
    __device__ __forceinline__ void issue(int k0, int K, unsigned lds0, int img, int dump) const {
        const u32x4 rs = stage_rsrc(k0, K);
#pragma unroll
        for (int i = 0; i < NP; ++i) issue_one(i, rs, lds0, img, dump);
    }
};

// MN-contiguous operand [k][mn] (a [k][n] weight): transpose-read image in ROW-MAJOR pieces.  A piece = 4 k rows x 128 columns =
// four 256-byte row segments, lane = 16 kr + pc requests 16-byte chunk pc ^ (4 kr) of row 4 kb + kr: sixteen consecutive lanes read
// ONE contiguous 256-byte segment (two whole cache lines).  (First form, measured: the sub-block-major image of gemm_bf16_core.h,
// where consecutive lane PAIRS jump to the next k row -- 32-byte fragments of four different lines per eight lanes -- ran the
// [k][n] shapes 7-16 us slower than the [n][k] ones.)  ds_read_b64_tr_b16 takes a per-lane address, so the 4 x 16 sub-block a
// transpose read assembles does not have to be contiguous: its row kr lives at kr * 256 + ((2 ib + half) ^ (4 kr)) * 16 inside the
// piece, and the XOR spreads the four rows of a sub-block -- which all start at the same bank in a plain row-major piece -- over
// the 256-byte bank row: the 32 lanes of a service group (2 sub-blocks x 4 rows x 4 eight-byte quarters) cover all 64 banks once.
// piece P = kb * NH + nh (kb: k group of 4, nh: 128-column half), wave w issues pieces w + 8 i.
template <int BMN>
struct RingDmaMN {
    static constexpr int NH = BMN / 128;
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
        const int kr = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int P = wave + 8 * i;
            const int kb = P / NH, nh = P % NH;
            const int col = mn0 + nh * 128 + 8 * (pc ^ (4 * kr));
            voff[i] = (col + 8 <= MN) ? (unsigned)(4 * kb + kr) * ld2b + 2u * (unsigned)col : BUF_OOB;
        }
    }
    __device__ __forceinline__ u32x4 stage_rsrc(int k0, int K) const {
        const unsigned left = (k0 < K) ? (unsigned)(text - (long long)k0 * ld2b) : 0u;
        return ring_rsrc(base + (long long)k0 * (ld2b >> 1), left);
    }
    __device__ __forceinline__ void issue_one(const int I, u32x4 rs, unsigned lds0, int img, int) const {
        ring_dma_piece(rs, lds0 + (unsigned)(img + (wave + 8 * I) * 1024), voff[I]);
    }
    __device__ __forceinline__ void issue(int k0, int K, unsigned lds0, int img, int dump) const {
        const u32x4 rs = stage_rsrc(k0, K);
#pragma unroll
        for (int i = 0; i < NP; ++i) issue_one(i, rs, lds0, img, dump);
    }
};

// lane-constant part of the transpose-read address of column block `col_base` (multiple of 32): lane = 16 g + t hands in the
// 8-byte chunk (row kr = t >> 2, quarter q = t & 3) of sub-block ib = col_base / 16 + (g & 1); k group 2 (g >> 1) (+ 1 for the second read)
template <int NH>
__device__ __forceinline__ int ring_tr_lane_off(int col_base, int lane) {
    const int g = lane >> 4, t = lane & 15;
    const int kr = t >> 2, q = t & 3;
    const int ibg = (col_base >> 4) + (g & 1);
    const int nh = ibg >> 3, ib = ibg & 7;
    return (2 * (g >> 1) * NH + nh) * 1024 + kr * 256 + (((2 * ib + (q >> 1)) ^ (4 * kr)) * 16) + (q & 1) * 8;
}
// MFMA fragment (8 consecutive k, starting at 16 kk + 8 (lane >> 5), of column col_base + (lane & 31)) out of the image
template <int NH>
__device__ __forceinline__ bf16x8 ring_frag_tr(const char *img, int lane_off, int kk) {
    const char *p = img + lane_off + kk * (4 * NH * 1024);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + NH * 1024));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// one 64-deep stage: NM (<= TM) row blocks of this wave x TN column blocks x 4 k-steps, with the DMA requests of the stage
// NS - 1 ahead spread over the k-steps.  First form (measured, profiles/r05_ring_ablation.txt): every wave issued its 7 requests
// right behind the barrier and multiplied afterwards -- all 8 waves of the workgroup are in the same phase, so nobody used the
// matrix pipe while the requests were issued (60-180 cycles each) and nobody issued while it ran: loop time = MFMA part + request
// part.  Now k-step kk carries the pieces [kk PW / 4, (kk + 1) PW / 4) in front of its MFMAs and the fragments of k-step kk + 1 are read
// before the MFMAs of kk (sched_group_barrier pins reads against MFMAs).
template <int IDX, int P0, int P1, int NPA_, class LA, class LB>
__device__ __forceinline__ void ring_issue_range(const LA &la, const LB &lb, u32x4 rsa, u32x4 rsb, unsigned lds0, int ia, int ib, int dump, int abl) {
    if constexpr (IDX < P1) {
        if constexpr (IDX >= P0) {
            if constexpr (IDX < NPA_) { if (!(abl & 2)) la.issue_one(IDX, rsa, lds0, ia, dump); }
            else { if (!(abl & 4)) lb.issue_one(IDX - NPA_, rsb, lds0, ib, dump); }
        }
        ring_issue_range<IDX + 1, P0, P1, NPA_>(la, lb, rsa, rsb, lds0, ia, ib, dump, abl);
    }
}
template <int NM, int TM, int TN, bool AKC, bool BKC, bool ABL, class LA, class LB>
__device__ __forceinline__ void ring_stage(const char *As, const char *Bs, f32x16 (&acc)[TM][TN], int a_lane, const int (&atr)[TM], int b_lane, const int (&btr)[TN],
                                           const int (&xo)[4], const LA &la, const LB &lb, int k_next, int K, unsigned lds0, int na, int nb, int dump,
                                           int abl) {
    const u32x4 rsa = la.stage_rsrc(k_next, K), rsb = lb.stage_rsrc(k_next, K);
    constexpr int NPA = TM, NPB = 2 * TN, PW = NPA + NPB;
    constexpr int READS = NM * (AKC ? 1 : 2) + TN * (BKC ? 1 : 2), MF = NM * TN;
    bf16x8 a[2][NM > 0 ? NM : 1], b[2][TN];
    auto read = [&](const int kk, const int set) {
#pragma unroll
        for (int mi = 0; mi < NM; ++mi) {
            if constexpr (AKC) a[set][mi] = *reinterpret_cast<const bf16x8 *>(As + a_lane + mi * (32 * RING_STAGE_ROW) + xo[kk]);
            else a[set][mi] = ring_frag_tr<TM / 2>(As, atr[mi], kk);
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            if constexpr (BKC) b[set][ni] = *reinterpret_cast<const bf16x8 *>(Bs + b_lane + ni * (32 * RING_STAGE_ROW) + xo[kk]);
            else b[set][ni] = ring_frag_tr<TN>(Bs, btr[ni], kk);
        }
    };
    if constexpr (NM == 0) {             // a row wave without rows in this tile: it still issues its share of the requests
        ring_issue_range<0, 0, PW, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, abl);
        return;
    } else {
        if constexpr (ABL) {             // timing experiments (DETR_HIP_RING_ABLATE): requests up front, under their switches; plain k-steps
            ring_issue_range<0, 0, PW, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, abl);
            if (abl & 1) return;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                read(kk, 0);
#pragma unroll
                for (int mi = 0; mi < NM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][mi], b[0][ni], acc[mi][ni], 0, 0, 0);
            }
            return;
        } else {
        read(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) read(kk + 1, (kk + 1) & 1);
            // the stage's requests go out during its first RING_ISSUE_STEPS k-steps (a two-stage ring waits for ALL of them at the top of the
            // next stage: what is requested in the last k-step has its whole latency exposed there -- profiles/r05_ring_trace.txt)
            constexpr int IS = DETR_RING_ISSUE_STEPS;
            if (kk == 0) ring_issue_range<0, (0 * PW) / IS, (1 * PW) / IS, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, 0);
            if (kk == 1 && IS >= 2) ring_issue_range<0, (1 * PW) / IS, (IS >= 2 ? (2 * PW) / IS : PW), NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, 0);
            if (kk == 2 && IS >= 3) ring_issue_range<0, (IS >= 3 ? (2 * PW) / IS : PW), (IS >= 3 ? (3 * PW) / IS : PW), NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, 0);
            if (kk == 3 && IS >= 4) ring_issue_range<0, (IS >= 4 ? (3 * PW) / IS : PW), PW, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, 0);
#pragma unroll
            for (int mi = 0; mi < NM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][mi], b[kk & 1][ni], acc[mi][ni], 0, 0, 0);
        }
        // pinned order: reads(0) reads(1) | MFMAs(0) + pieces | reads(2) | MFMAs(1) + pieces | reads(3) | MFMAs(2) + pieces | MFMAs(3) + pieces
        // (the requests are inline assembly: `asm volatile` keeps them in source order between the k-steps, sched_group_barrier cannot see them)
        sgb_ds_read<2 * READS>();
        sgb_mfma_block<MF, 0, 0, 0>();
        sgb_ds_read<READS>();
        sgb_mfma_block<MF, 0, 0, 0>();
        sgb_ds_read<READS>();
        sgb_mfma_block<MF, 0, 0, 0>();
        sgb_mfma_block<MF, 0, 0, 0>();
        }
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
    RingDmaK<NPA, 2, DETR_RING_NT != 0> la;
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
    const int dump = ra.dump_off;
    const unsigned lds0 = ring_lds_addr(ring_smem);
    auto stage_a = [&](int s) { return s * ra.stage_bytes; };                  // byte offsets into ring_smem
    auto stage_b = [&](int s) { return s * ra.stage_bytes + a_bytes; };
    // fragment addressing: row (base multiple of 32) + (lane & 31) -> its swizzle term is ((lane & 31) >> 1) & 7 whatever the block
    const int l31 = lane & 31, h = lane >> 5;
    const int sw = (l31 >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xo[kk] = 16 * ((2 * kk + h) ^ sw);
    const int a_lane = (wm * T::WTM + l31) * RING_STAGE_ROW;
    const int b_lane = (wn * T::WTN + l31) * RING_STAGE_ROW;
    int atr_unused[TM] = {};
    int btr[TN];                                        // [k][n] weights: lane-constant transpose-read offsets of this wave's column blocks
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) btr[ni] = ring_tr_lane_off<TN>(wn * T::WTN + ni * 32, lane);

    const int nkt = g.K / RING_BK;
    const int abl = ra.ablate;                          // (kernel argument: uniform)
#if (DETR_ABLATE & 64) != 0
    long long *trl = reinterpret_cast<long long *>(ring_smem + ra.trace_off);
    const int tr_sel = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1));
    const bool tr_w = tr_sel >= 0 && wave == 0;
    RING_STAMP(RING_TR_STEPS * 3 + 0);
    if (tr_w && lane == 0) trl[RING_TR_STEPS * 3 + 4] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
        if (!(abl & 2)) la.issue(t * RING_BK, g.K, lds0, stage_a(t), dump);
        if (!(abl & 4)) lb.issue(t * RING_BK, g.K, lds0, stage_b(t), dump);
    }
    // The K loop exists once per block count NM a wave can have (chosen ONCE, outside the loop: a per-stage switch made the
    // compiler shuffle all accumulators between the branches' register assignments every iteration)
    auto kloop_ab = [&](auto NMC, auto ABLC) {
        constexpr int NM = decltype(NMC)::value;
        constexpr bool ABL = decltype(ABLC)::value;
        int cur = 0, nxt = NS - 1;                      // ring slots of stage t and of stage t + NS - 1
        for (int t = 0; t < ((abl & 16) ? 0 : nkt); ++t) {
            RING_STAMP_STEP(t, 0);
            ring_wait_vmcnt<(NS - 2) * PW>();
            RING_STAMP_STEP(t, 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            RING_STAMP_STEP(t, 2);
            ring_stage<NM, TM, TN, true, BKC, ABL>(ring_smem + stage_a(cur), ring_smem + stage_b(cur), acc, a_lane, atr_unused, b_lane, btr, xo, la, lb,
                                             (t + NS - 1) * RING_BK, g.K, lds0, stage_a(nxt), stage_b(nxt), dump, abl);
            __builtin_amdgcn_sched_barrier(0);
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
    };
    // (the ablation form of the loop is a separate loop as well: a run-time switch inside the loop body costs the accumulator shuffle again)
    auto kloop = [&](auto NMC) {
        if (abl) kloop_ab(NMC, std::integral_constant<bool, true>{});
        else kloop_ab(NMC, std::integral_constant<bool, false>{});
    };
    RING_STAMP(RING_TR_STEPS * 3 + 1);
    if (nmi == TM) kloop(std::integral_constant<int, TM>{});
    else if (TM >= 2 && nmi == TM - 1) kloop(std::integral_constant<int, (TM >= 2 ? TM - 1 : 0)>{});
    else if (TM >= 3 && nmi == TM - 2) kloop(std::integral_constant<int, (TM >= 3 ? TM - 2 : 0)>{});
    else if (TM >= 4 && nmi == TM - 3) kloop(std::integral_constant<int, (TM >= 4 ? TM - 3 : 0)>{});
    else kloop(std::integral_constant<int, 0>{});
    RING_STAMP(RING_TR_STEPS * 3 + 2);
    ring_wait_vmcnt<0>();                               // the trailing empty requests still write (zeros) into the ring:
    __syncthreads();                                    // nobody reuses the array (epilogue staging) before they have landed
    if (abl & 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) ablate_keep(acc[i][j]);
        return;
    }
    epilogue<BM, BN, WGM, WGN>(acc, reinterpret_cast<float *>(ring_smem), g.C, g.ldc, row_end, g.N, m0, n0, wm, wn, lane, wave, g.e);
#if (DETR_ABLATE & 64) != 0
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    RING_STAMP(RING_TR_STEPS * 3 + 3);
    if (tr_w && lane == 0) trl[RING_TR_STEPS * 3 + 5] = (long long)__builtin_amdgcn_s_memrealtime();
    if (tr_w) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < RING_TR_N; i += 64) ring_trace[tr_sel][i] = trl[i];
    }
#endif
}

template <int TM, int TN, bool BKC, int NS>
__global__ __launch_bounds__(RING_THREADS) void gemm_ring_kernel(RingArgs ra) {
    // (device pass only: the host pass of hipcc needs the kernel's symbol, not its body -- and rejects the body's device-function templates
    //  with a bare "substitution failure" once the loaders are passed down as template arguments)
#if defined(__HIP_DEVICE_COMPILE__)
    gemm_ring_body<TM, TN, BKC, NS>(ra, xcd_remap((int)blockIdx.x, (int)gridDim.x));
#endif
}

// ===========================================================================================================================
// fp32 form of the ring (the exact-f32 parity mode: v_mfma_f32_32x32x2_f32, the reference's own arithmetic).  Same pipeline, same
// tile grid; a stage is 32 k deep (128-byte rows again).  fp32 GEMMs are MFMA-bound (the f32 MFMA runs at 1/16 of the bf16 rate),
// and the 4-wave engine sat at 0.50 of that peak: one 32x32 accumulator per wave -- eight dependent MFMAs per K tile back to back,
// issue stalls 52 % of the cycles (NOTEBOOK 7c).  Here a wave owns TM x TN independent accumulators and consecutive MFMAs never
// touch the same one.
//   * K-contiguous operand ([rows][k]): the same XOR-swizzled 128-byte-row image; a lane reads the 16-byte chunk of 4 consecutive k of
//     its row ONCE per two MFMA steps (both lane halves the same address: broadcast) and picks element 2 j + h for step j -- the
//     operand map of the 32x32x2 MFMA is lane l -> (row l & 31, k = l >> 5);
//   * MN-contiguous operand ([k][mn]): the image is the operand's own orientation, one 4 BMN-byte row per k; a fragment is one
//     ds_read_b32 per MFMA step (lanes 0-31 on consecutive dwords of row 2 s, lanes 32-63 of row 2 s + 1: conflict-free, no
//     transposition -- the f32 MFMA takes one value per lane).
// Every accumulator sees k in ascending pairs (2 s, 2 s + 1) like gemm_f32_body: bit-identical results.
template <int BMN>
struct RingDmaMN32 {       // [k][mn] fp32: piece = 1 KB = 256 floats = 256 / BMN rows of the stage image; lane: row piece * RPP + lane / LPR, columns 4 (lane % LPR) ..
    static constexpr int LPR = BMN / 4;                  // lanes per image row
    static constexpr int RPP = 64 / LPR;                 // image rows per piece
    static constexpr int NP = (32 / RPP) / 8;            // pieces per wave and stage (32 k rows)
    static constexpr int NPIECES = NP;
    static_assert(BMN == 128 || BMN == 256, "128 or 256 columns");
    const char *base;
    long long text;
    unsigned ld4b;
    unsigned voff[NP];
    int wave;
    __device__ __forceinline__ void init(const void *p, long long ld, int mn0, int MN, int K, int lane, int wave_) {
        wave = wave_;
        base = reinterpret_cast<const char *>(p);
        text = ((long long)(K - 1) * ld + MN) * 4;
        ld4b = (unsigned)(ld * 4);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int P = wave + 8 * i;
            const int k = P * RPP + lane / LPR;
            const int col = mn0 + 4 * (lane % LPR);
            voff[i] = (col + 4 <= MN) ? (unsigned)k * ld4b + 4u * (unsigned)col : BUF_OOB;
        }
    }
    __device__ __forceinline__ u32x4 stage_rsrc(int k0, int K) const {
        const unsigned left = (k0 < K) ? (unsigned)(text - (long long)k0 * ld4b) : 0u;
        return ring_rsrc(base + (long long)k0 * ld4b, left);
    }
    __device__ __forceinline__ void issue_one(const int I, u32x4 rs, unsigned lds0, int img, int) const {
        ring_dma_piece(rs, lds0 + (unsigned)(img + (wave + 8 * I) * 1024), voff[I]);
    }
    __device__ __forceinline__ void issue(int k0, int K, unsigned lds0, int img, int dump) const {
        const u32x4 rs = stage_rsrc(k0, K);
#pragma unroll
        for (int i = 0; i < NP; ++i) issue_one(i, rs, lds0, img, dump);
    }
};

// one 32-deep fp32 stage: NM x TN accumulators x 16 MFMA steps; the requests of the stage NS - 1 ahead ride between the k groups
template <int NM, int TM, int TN, bool AKC, bool BKC, int BM, int BN, bool ABL, class LA, class LB>
__device__ __forceinline__ void ring_stage_f32(const char *As, const char *Bs, f32x16 (&acc)[TM][TN], int a_lane, int b_lane, const int (&xo8)[8],
                                               int h, const LA &la, const LB &lb, int k_next, int K, unsigned lds0, int na, int nb, int dump, int abl) {
    constexpr int NPA = LA::NPIECES, NPB = LB::NPIECES, PW = NPA + NPB;
    const u32x4 rsa = la.stage_rsrc(k_next, K), rsb = lb.stage_rsrc(k_next, K);
    if constexpr (NM == 0) {
        ring_issue_range<0, 0, PW, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, ABL ? abl : 0);
        return;
    } else {
        if constexpr (ABL) {             // timing experiments: requests up front under their switches; bit 0: no fragment reads / MFMAs
            ring_issue_range<0, 0, PW, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, abl);
            if (abl & 1) return;
        }
#pragma unroll
        for (int gq = 0; gq < 8; ++gq) {                 // k group of 4: two MFMA steps
            float av[2][NM], bv[2][TN];
#pragma unroll
            for (int mi = 0; mi < NM; ++mi) {
                if constexpr (AKC) {
                    const float4 v = *reinterpret_cast<const float4 *>(As + a_lane + mi * (32 * RING_STAGE_ROW) + xo8[gq]);
                    av[0][mi] = h ? v.y : v.x;
                    av[1][mi] = h ? v.w : v.z;
                } else {
                    av[0][mi] = *reinterpret_cast<const float *>(As + a_lane + mi * 128 + (4 * gq) * (BM * 4));
                    av[1][mi] = *reinterpret_cast<const float *>(As + a_lane + mi * 128 + (4 * gq + 2) * (BM * 4));
                }
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                if constexpr (BKC) {
                    const float4 v = *reinterpret_cast<const float4 *>(Bs + b_lane + ni * (32 * RING_STAGE_ROW) + xo8[gq]);
                    bv[0][ni] = h ? v.y : v.x;
                    bv[1][ni] = h ? v.w : v.z;
                } else {
                    bv[0][ni] = *reinterpret_cast<const float *>(Bs + b_lane + ni * 128 + (4 * gq) * (BN * 4));
                    bv[1][ni] = *reinterpret_cast<const float *>(Bs + b_lane + ni * 128 + (4 * gq + 2) * (BN * 4));
                }
            }
            if constexpr (!ABL) {
                if (gq == 1) ring_issue_range<0, (0 * PW) / 4, (1 * PW) / 4, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, 0);
                if (gq == 3) ring_issue_range<0, (1 * PW) / 4, (2 * PW) / 4, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, 0);
                if (gq == 5) ring_issue_range<0, (2 * PW) / 4, (3 * PW) / 4, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, 0);
                if (gq == 7) ring_issue_range<0, (3 * PW) / 4, (4 * PW) / 4, NPA>(la, lb, rsa, rsb, lds0, na, nb, dump, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int mi = 0; mi < NM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][mi], bv[j][ni], acc[mi][ni], 0, 0, 0);
        }
    }
}

// unsplit fp32 GEMM: A [M][K] (K-contiguous), B [N][K] or [K][N]; row-balanced tile grid as in gemm_ring_body
template <int TM, int TN, bool BKC, int NS>
__device__ __forceinline__ void gemm_ring_f32_body(const RingArgs &ra, const int id) {
    constexpr int BM = 64 * TM, BN = 128 * TN, WGM = 2, WGN = 4, BKF = 32;
    using T = TileCfg<BM, BN, WGM, WGN>;
    extern __shared__ __attribute__((aligned(1024))) char ring_smem[];
    const GemmArgs &g = ra.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tn = id % g.tiles_n, tm = id / g.tiles_n;
    const int m0 = tm * ra.tile_rows, n0 = tn * BN;
    const int row_end = min(g.M, m0 + ra.tile_rows);
    const int my_rows = row_end - (m0 + wm * T::WTM);
    const int nmi = my_rows <= 0 ? 0 : (my_rows >= T::WTM ? TM : (my_rows + 31) >> 5);
    using LAt = RingDmaK<TM, 4>;
    LAt la;
    la.init(g.A, g.lda, m0, row_end, g.M, g.K, ra.a_rows8, lane, wave);
    using LB = typename std::conditional<BKC, RingDmaK<2 * TN, 4>, RingDmaMN32<BN>>::type;
    LB lb;
    if constexpr (BKC) lb.init(g.B, g.ldb, n0, g.N, g.N, g.K, BN, lane, wave);
    else lb.init(g.B, g.ldb, n0, g.N, g.K, lane, wave);
    constexpr int PW = LAt::NPIECES + LB::NPIECES;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int a_bytes = ra.a_rows8 * RING_STAGE_ROW;
    const int dump = ra.dump_off;
    const unsigned lds0 = ring_lds_addr(ring_smem);
    auto stage_a = [&](int s) { return s * ra.stage_bytes; };
    auto stage_b = [&](int s) { return s * ra.stage_bytes + a_bytes; };
    const int l31 = lane & 31, h = lane >> 5;
    const int sw = (l31 >> 1) & 7;
    int xo8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) xo8[q] = 16 * (q ^ sw);
    const int a_lane = (wm * T::WTM + l31) * RING_STAGE_ROW;
    // [n][k]: row of the column block; [k][n]: dword (wn * WTN + l31) of image row h (the MFMA's k = 2 s + h)
    const int b_lane = BKC ? (wn * T::WTN + l31) * RING_STAGE_ROW : (wn * T::WTN + l31) * 4 + h * (BN * 4);
    const int nkt = g.K / BKF;
    const int abl = ra.ablate;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
        if (!(abl & 2)) la.issue(t * BKF, g.K, lds0, stage_a(t), dump);
        if (!(abl & 4)) lb.issue(t * BKF, g.K, lds0, stage_b(t), dump);
    }
    auto kloop_ab = [&](auto NMC, auto ABLC) {
        constexpr int NM = decltype(NMC)::value;
        constexpr bool ABL = decltype(ABLC)::value;
        int cur = 0, nxt = NS - 1;
        for (int t = 0; t < ((abl & 16) ? 0 : nkt); ++t) {
            ring_wait_vmcnt<(NS - 2) * PW>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ring_stage_f32<NM, TM, TN, true, BKC, BM, BN, ABL>(ring_smem + stage_a(cur), ring_smem + stage_b(cur), acc, a_lane, b_lane, xo8, h, la, lb,
                                                               (t + NS - 1) * BKF, g.K, lds0, stage_a(nxt), stage_b(nxt), dump, abl);
            __builtin_amdgcn_sched_barrier(0);
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
    };
    auto kloop = [&](auto NMC) {
        if (abl) kloop_ab(NMC, std::integral_constant<bool, true>{});
        else kloop_ab(NMC, std::integral_constant<bool, false>{});
    };
    if (nmi == TM) kloop(std::integral_constant<int, TM>{});
    else if (TM >= 2 && nmi == TM - 1) kloop(std::integral_constant<int, (TM >= 2 ? TM - 1 : 0)>{});
    else if (TM >= 3 && nmi == TM - 2) kloop(std::integral_constant<int, (TM >= 3 ? TM - 2 : 0)>{});
    else if (TM >= 4 && nmi == TM - 3) kloop(std::integral_constant<int, (TM >= 4 ? TM - 3 : 0)>{});
    else kloop(std::integral_constant<int, 0>{});
    ring_wait_vmcnt<0>();
    __syncthreads();
    if (abl & 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) ablate_keep(acc[i][j]);
        return;
    }
    epilogue<BM, BN, WGM, WGN, false>(acc, reinterpret_cast<float *>(ring_smem), g.C, g.ldc, row_end, g.N, m0, n0, wm, wn, lane, wave, g.e);
}

template <int TM, int TN, bool BKC, int NS>
__global__ __launch_bounds__(RING_THREADS) void gemm_ring_f32_kernel(RingArgs ra) {
#if defined(__HIP_DEVICE_COMPILE__)
    gemm_ring_f32_body<TM, TN, BKC, NS>(ra, xcd_remap((int)blockIdx.x, (int)gridDim.x));
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// Split-K weight gradients on the ring: C[M, N] partial = sum over this split's k of A[k][m] * B[k][n], both operands bf16 and
// MN-contiguous (the 1x1-convolution weight gradients of the backbone: x^T dy over B*H*W pixels).  Same tile (64 TM x 128 TN), same
// split ranges, same MFMA order per accumulator and the same tile-ordered slabs as gemm_bf16c_k64_kernel<BM, BN, 2, 2, false, false>
// (bit-identical; the reduce launch is unchanged) -- but 8 waves per workgroup instead of 4 (the split-K launches run ONE
// workgroup per CU: the 4-wave engine leaves one wave per SIMD, whose K tile cost ~2500 cycles for 512 of MFMA, NOTEBOOK 4b), both
// operands by LDS-DMA into transpose-read images, NS - 1 stages in flight.
// The 2 x 4 wave grid stores its accumulator blocks at the slab positions a 2 x 2 grid would use (gemm_core.h slab_ts_unit):
// block (wm, wn, mi, ni) is block (w' = 2 wm + (wn >> 1), mi, ni' = (wn & 1) TN + ni) of the 2 x 2 layout with TN' = 2 TN.
template <int TM, int TN, int NS>
__device__ __forceinline__ void gemm_ring_wgrad_body(const RingArgs &ra) {
    constexpr int BM = 64 * TM, BN = 128 * TN, WGM = 2, WGN = 4;
    static_assert(TM % 2 == 0, "the A image is made of 128-row pieces");
    using T = TileCfg<BM, BN, WGM, WGN>;
    extern __shared__ __attribute__((aligned(1024))) char ring_smem[];
    const GemmArgs &g = ra.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    // (split, tile): the whole list remapped over the XCDs so that the tiles of a split share one L2 (gemm_kernels.h gemm_work_item)
    const int tiles = g.tiles_m * g.tiles_n;
    const int id = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tile = id % tiles, split = id / tiles;
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nkt32 = (g.K + BF_BK - 1) / BF_BK;
    const int per = (nkt32 + g.split_k - 1) / g.split_k;
    const int kbeg = split * per * BF_BK;
    const int kend = min(g.K, kbeg + per * BF_BK);
    if (kbeg >= kend) return;
    RingDmaMN<BM> la;
    RingDmaMN<BN> lb;
    la.init(g.A, g.lda, m0, g.M, kend, lane, wave);
    lb.init(g.B, g.ldb, n0, g.N, kend, lane, wave);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    constexpr int A_BYTES = BM * RING_STAGE_ROW, STAGE = (BM + BN) * RING_STAGE_ROW, PW = TM + 2 * TN;
    const unsigned lds0 = ring_lds_addr(ring_smem);
    int atr[TM], btr[TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) atr[mi] = ring_tr_lane_off<TM / 2>(wm * T::WTM + mi * 32, lane);
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) btr[ni] = ring_tr_lane_off<TN>(wn * T::WTN + ni * 32, lane);
    const int xo[4] = {0, 0, 0, 0};
    const int nkt = (kend - kbeg + RING_BK - 1) / RING_BK;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
        la.issue(kbeg + t * RING_BK, kend, lds0, t * STAGE, 0);
        lb.issue(kbeg + t * RING_BK, kend, lds0, t * STAGE + A_BYTES, 0);
    }
    const int abl = ra.ablate;
    auto kloop = [&](auto ABLC) {
        constexpr bool ABL = decltype(ABLC)::value;
        int cur = 0, nxt = NS - 1;
        for (int t = 0; t < nkt; ++t) {
            ring_wait_vmcnt<(NS - 2) * PW>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ring_stage<TM, TM, TN, false, false, ABL>(ring_smem + cur * STAGE, ring_smem + cur * STAGE + A_BYTES, acc, 0, atr, 0, btr, xo, la, lb,
                                                      kbeg + (t + NS - 1) * RING_BK, kend, lds0, nxt * STAGE, nxt * STAGE + A_BYTES, 0, abl);
            __builtin_amdgcn_sched_barrier(0);
            cur = (cur + 1 == NS) ? 0 : cur + 1;
            nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
    };
    if (abl) kloop(std::integral_constant<bool, true>{});      // (DETR_HIP_RING_ABLATE timing experiments: bits 1, 2, 4 as in gemm_ring_body)
    else kloop(std::integral_constant<bool, false>{});
    ring_wait_vmcnt<0>();
    // tile-ordered slab of this (split, tile), in the 2 x 2 grid's unit order
    float4 *slab = reinterpret_cast<float4 *>(g.C + (long long)split * g.part_stride + (long long)tile * (BM * BN));
    const int w2 = 2 * wm + (wn >> 1);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int blk = (w2 * TM + mi) * (2 * TN) + (wn & 1) * TN + ni;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                slab[(blk * 4 + q) * 64 + lane] = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]);
        }
}

template <int TM, int TN, int NS>
__global__ __launch_bounds__(RING_THREADS) void gemm_ring_wgrad_kernel(RingArgs ra) {
#if defined(__HIP_DEVICE_COMPILE__)
    gemm_ring_wgrad_body<TM, TN, NS>(ra);
#endif
}

}  // namespace detr
