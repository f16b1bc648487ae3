// conv_halo_dma.h -- the halo-staged stride-1 3x3 convolution (conv_halo.h) with BOTH operands filled by LDS-DMA (round 5).
// Same tile (4 x 32 output pixels x 128 channels), same 2 x 2 wave grid, same MFMA order per accumulator and the same epilogue as
// conv3x3_halo_bf16_kernel<128, DGRAD, 4>: results are bit-identical.  What changes is who moves the operands:
//
// Ablation builds of the register-staged kernel (scripts/experiments/ablate.sh, profiles/r05_conv_ablation.txt), 50x84x256 forward:
// 66.7 us complete | 42.5 without MFMAs | 43.2 without the operand requests and their LDS stores | 44.4 without fragment reads |
// 37.5 with the MFMAs alone | 12.4 with an empty loop -- three phases of ~23 us each that do not overlap, three workgroups per CU or not:
// a wave that waits for its 16-byte loads, writes them to LDS and waits at the barrier is not issuing MFMAs, and its neighbours are in the
// same step.  Here (the lesson of the ring GEMM, gemm_ring.h) the waves hold no staging registers and write nothing to LDS: per step a
// wave issues two or three 1 KB DMA pieces (inline assembly: the compiler must not know, it would wait for vmcnt(0) in front of the
// transpose reads) and otherwise reads fragments and multiplies.
//   * input patch of a 32-channel chunk: [pixel][32 ch] in UNPADDED 64-byte rows, the 16-byte chunk c of patch pixel p at position
//     c ^ ((p >> 2) & 3): the 16 lanes of a ds_read_b128 service group ({0-3, 12-15, 20-27} ...) read patch pixels that are distinct
//     mod 16 (any tap shift) and hit 16 distinct 16-byte slots.  A piece = 16 patch pixels; 13 pieces (204 pixels), the three spare piece
//     slots of the four waves go to a dump area.  The patch of chunk c + 1 is requested during taps 0..3 of chunk c (its buffer was last
//     read in chunk c - 1);
//   * kernel tile of a step: forward [k = ci][n = co] as the row-major transpose-read image of gemm_ring.h (4 k x 128 n pieces), input
//     gradient [n = ci][k = co] as 64-byte rows with the same XOR swizzle (16 rows per piece); a ring of THREE tiles -- the tile of step s
//     is requested during step s - 2; 9 taps = 0 mod 3, so the ring slot of a step is its tap index mod 3: static.
//   * per step: s_waitcnt vmcnt(pieces of the previous step) -> s_barrier -> fragments + 8 MFMAs, the requests between the two k-steps'
//     MFMAs (LATE; the guide prices a DMA piece at 100-185 cycles of issue inside a phase that is also reading fragments, 25-60 in a
//     later gap).  One barrier per step (the register-staged kernel: one as well), no LDS store pass, no vmcnt(0).
// LDS: 2 x 13 KB patch + 3 x 8 KB kernel tiles + 1 KB dump = 51 KB: three workgroups per CU as before.
//
// Measured (scripts/micro_conv.py, profiles/r05_micro_conv_dma.txt; us forward / input gradient, register-staged -> this kernel):
// 100x167x128 53.2 / 54.0 -> 49.6 / 50.9, 50x84x256 54.7 / 55.3 -> 49.9 / 51.9, 25x42x512 66.0 / 68.9 -> 61.3 / 62.3: -5 ... -10 %, not
// the -25 % the ablation promised.  A per-step timeline of the kernel (s_memtime stamps of one wave, -DDETR_ABLATE=64,
// scripts/experiments/conv_trace.py, profiles/r05_conv_trace.txt) shows why: the counted wait and the barrier cost nothing beyond the
// stamp itself (116-130 ticks each) -- the operands are always there -- and all of a step's 650-1500 ticks are its "work" section:
// ~85 instructions of one wave around 8 MFMAs (256 cycles), three such waves per SIMD, 9.5 KB of DMA per 32 MFMAs of a workgroup
// (437 MB per launch through L2 -> LDS, 8 TB/s).  Two variants that attack latency instead were built, measured and removed
// (profiles/r05_ab_results.txt): MFMAs one step BEHIND the fragment reads (two fragment sets; 52.9 vs 52.7 us) and a six-slot tile ring at two
// workgroups per CU (tiles requested five steps ahead; 59.8 vs 51.5 us, the second round of a 624-workgroup grid on 512 slots).  What would
// help is fewer kernel-tile bytes per MFMA (a 256-pixel tile halves them) without the tile-count losses of 8-row tiles (conv_halo.h) --
// NOTEBOOK.md section 7a.
#pragma once
#include "conv_halo.h"
#include "gemm_ring.h"

namespace detr {

constexpr int CHD_TH = 4, CHD_BN = 128, CHD_NB = 3;
constexpr int CHD_PIX = (CHD_TH + 2) * CH_PW;            // 204 patch pixels
constexpr int CHD_PP = (CHD_PIX + 15) / 16;              // 13 pieces of 16 pixels
constexpr int CHD_PATCH = CHD_PP * 1024;                 // bytes of one patch buffer
constexpr int CHD_BT = CHD_BN * BF_BK * 2;               // bytes of one kernel tile (8 KB)
constexpr int CHD_OFF_B = 2 * CHD_PATCH;
constexpr int CHD_OFF_DUMP = CHD_OFF_B + CHD_NB * CHD_BT;
constexpr int CHD_RING = CHD_OFF_DUMP + 1024;
constexpr int CHD_EPI = StageCfg<CHD_BN, 2>::BYTES;
constexpr int CHD_SMEM = CHD_RING > CHD_EPI ? CHD_RING : CHD_EPI;

#if (DETR_ABLATE & 64) != 0
// timeline experiment (scripts/experiments/conv_trace.py): wave 0 of three workgroups stamps the cycle counter in front of the counted wait,
// in front of the barrier and behind it, every step; [workgroup][40 steps x 3 + kernel start, loop start, loop end, kernel end, and the
// 100 MHz real-time counter at kernel start / end]
constexpr int CHD_TR_STEPS = 40, CHD_TR_N = CHD_TR_STEPS * 3 + 6;
__device__ long long chd_trace[3][CHD_TR_N];
#define CHD_STAMP(idx) do { if (tr_w) { const long long t_ = __builtin_readcyclecounter(); if (lane == 0) trl[(idx)] = t_; } } while (0)
#define CHD_STAMP_STEP(s, k) do { if ((s) < CHD_TR_STEPS) CHD_STAMP((s) * 3 + (k)); } while (0)
#else
#define CHD_STAMP(idx) do { } while (0)
#define CHD_STAMP_STEP(s, k) do { } while (0)
#endif

template <bool DGRAD, bool LATE>
__global__ __launch_bounds__(GEMM_THREADS, 3) void conv3x3_halo_dma_bf16_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BN = CHD_BN, TH = CHD_TH, NB = CHD_NB;
    using T = TileCfg<TH * CH_TW, BN, 2, 2>;
    extern __shared__ __attribute__((aligned(1024))) char chd_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = id % a.tiles_n;
    int t = id / a.tiles_n;
    const int twi = t % a.Wp;
    t /= a.Wp;
    const int thi = t % a.Hp;
    const int n = t / a.Hp;
    const int h0 = thi * TH, w0 = twi * CH_TW, n0 = tn * BN;
    const int nchunks = a.Cs / BF_BK;
    const unsigned lds0 = ring_lds_addr(chd_smem);
#if (DETR_ABLATE & 64) != 0
    long long *trl = reinterpret_cast<long long *>(chd_smem + CHD_SMEM);
    const int tr_sel = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1));
    const bool tr_w = tr_sel >= 0 && wave == 0;
    CHD_STAMP(CHD_TR_STEPS * 3 + 0);
    if (tr_w && lane == 0) trl[CHD_TR_STEPS * 3 + 4] = (long long)__builtin_amdgcn_s_memrealtime();
#endif

    // ---- patch pieces of this wave: P = wave + 4 i (i < 4); lane -> patch pixel 16 P + lane / 4, slot position lane & 3
    const unsigned long long src_bytes = (unsigned long long)a.N * a.Hs * a.Ws * a.Cs * 2ull;
    unsigned pvoff[4];
    int plds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int P = wave + 4 * i;
        const int px = 16 * P + (lane >> 2);
        const int pr = px / CH_PW, pc = px - pr * CH_PW;
        const int h = h0 - 1 + pr, w = w0 - 1 + pc;
        const int c = (lane & 3) ^ ((px >> 2) & 3);                  // source chunk that belongs at this lane's slot
        const bool ok = px < CHD_PIX && h >= 0 && h < a.Hs && w >= 0 && w < a.Ws;
        pvoff[i] = ok ? ((unsigned)((n * a.Hs + h) * a.Ws + w) * (unsigned)a.Cs) * 2u + 16u * (unsigned)c : BUF_OOB;
        plds[i] = P < CHD_PP ? P * 1024 : -1;                        // (wave-uniform) spare slots: dump area
    }
    // piece i of the patch of chunk c into buffer c & 1 (chunk >= nchunks: empty descriptor, no traffic)
    auto patch_issue = [&](const int i, const int c) {
        const unsigned left = (c < nchunks) ? (unsigned)(src_bytes - (unsigned long long)c * (BF_BK * 2)) : 0u;
        const u32x4 rs = ring_rsrc(reinterpret_cast<const char *>(a.src) + (long long)c * (BF_BK * 2), left);
        const int base = (c & 1) * CHD_PATCH;
        const int in = plds[i] >= 0 ? -1 : 0;
        const int off = CHD_OFF_DUMP + (in & (base + plds[i] - CHD_OFF_DUMP));
        ring_dma_piece(rs, lds0 + (unsigned)off, pvoff[i]);
    };
    // ---- kernel-tile pieces of this wave: P = wave + 4 i (i < 2)
    const long long tapstride = (long long)a.Ci * a.Co;
    const unsigned long long w_bytes = 9ull * (unsigned long long)tapstride * 2ull;
    unsigned bvoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int P = wave + 4 * i;
        if constexpr (!DGRAD) {            // [k = ci][n = co]: piece = k group P (4 k rows x 128 n), lane = 16 kr + pc
            const int kr = lane >> 4, pc = lane & 15;
            const int col = n0 + 8 * (pc ^ (4 * kr));
            bvoff[i] = (col + 8 <= a.Cd) ? (unsigned)(4 * P + kr) * (unsigned)a.Co * 2u + 2u * (unsigned)col : BUF_OOB;
        } else {                           // [n = ci][k = co]: piece = rows 16 P .. 16 P + 15, lane -> row 16 P + lane / 4, slot position lane & 3
            const int r = 16 * P + (lane >> 2);
            const int c = (lane & 3) ^ ((r >> 2) & 3);
            bvoff[i] = (n0 + r < a.Cd) ? (unsigned)(n0 + r) * (unsigned)a.Co * 2u + 16u * (unsigned)c : BUF_OOB;
        }
    }
    // kernel tile of step (chunk c, patch offset tp) into ring slot `slot`
    auto b_issue = [&](const int c, const int tp, const int slot) {
        const int wt = DGRAD ? 8 - tp : tp;                          // input gradient: the kernel tap is the flipped offset
        // element offset of the tile's first row: forward rows k = ci (c * 32 ..), input gradient columns k = co (c * 32 ..)
        const long long e0 = wt * tapstride + (DGRAD ? (long long)c * BF_BK : (long long)c * BF_BK * a.Co);
        const unsigned left = (c < nchunks) ? (unsigned)(w_bytes - (unsigned long long)e0 * 2ull) : 0u;
        const u32x4 rs = ring_rsrc(reinterpret_cast<const char *>(a.w) + e0 * 2, left);
#pragma unroll
        for (int i = 0; i < 2; ++i) ring_dma_piece(rs, lds0 + (unsigned)(CHD_OFF_B + slot * CHD_BT + (wave + 4 * i) * 1024), bvoff[i]);
    };

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int l31 = lane & 31, hh = lane >> 5;
    int p0[T::TM];                                       // patch pixel of this lane's row in tile row (2 wm + mi), before the tap offset
#pragma unroll
    for (int mi = 0; mi < T::TM; ++mi) p0[mi] = (HaloGeom<TH>::RB * wm + mi) * CH_PW + l31;
    int btr[T::TN], brow[T::TN];
#pragma unroll
    for (int ni = 0; ni < T::TN; ++ni) {
        btr[ni] = ring_tr_lane_off<1>(wn * T::WTN + ni * 32, lane);                            // forward: transpose-read image
        const int r = wn * T::WTN + ni * 32 + l31;                                             // input gradient: row r, chunk (2 kk + hh) ^ ((r >> 2) & 3)
        brow[ni] = r * 64 + 16 * (hh ^ ((r >> 2) & 3));
    }
    bf16x8 fa[2][T::TM], fb[2][T::TN];                   // both k-steps' fragments are requested before the first MFMA
    auto read_tap = [&](const int pbuf, const int slot, const int tp) {
        const int toff = (tp / 3) * CH_PW + (tp - 3 * (tp / 3));
        const char *Pb = chd_smem + pbuf * CHD_PATCH;
        const char *Bs = chd_smem + CHD_OFF_B + slot * CHD_BT;
        int pa[T::TM];
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi) {
            int q = p0[mi];
            asm volatile("" : "+v"(q));                  // opaque: the nine taps' addresses are NOT loop invariants to keep in 18 registers
            const int p = q + toff;
            pa[mi] = p * 64 + 16 * (hh ^ ((p >> 2) & 3));            // k-step 0; k-step 1 is the chunk two further: ^ 32 bytes
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi) {
                if constexpr ((DETR_ABLATE & 16) != 0) fa[kk][mi] = __builtin_bit_cast(bf16x8, make_uint4(lane, mi, kk, 0x3f803f80u));
                else fa[kk][mi] = *reinterpret_cast<const bf16x8 *>(Pb + (pa[mi] ^ (32 * kk)));
            }
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) {
                if constexpr ((DETR_ABLATE & 16) != 0) fb[kk][ni] = __builtin_bit_cast(bf16x8, make_uint4(lane, ni, kk, 0x3f803f80u));
                else if constexpr (!DGRAD) fb[kk][ni] = ring_frag_tr<1>(Bs, btr[ni], kk);
                else fb[kk][ni] = *reinterpret_cast<const bf16x8 *>(Bs + (brow[ni] ^ (32 * kk)));
            }
        }
    };
    auto mma_half = [&](const int kk) {
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) {
                if constexpr ((DETR_ABLATE & 1) != 0) { ablate_keep(fa[kk][mi]); ablate_keep(fb[kk][ni]); }
                else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][mi], fb[kk][ni], acc[mi][ni], 0, 0, 0);
            }
    };

    // ---- pipeline.  Step s = 9 c + tp reads patch buffer c & 1 and kernel-tile slot tp % 3; during it the wave requests the tile of step
    // s + 2 (2 pieces) and, for tp < 4, piece tp of the patch of chunk c + 1 (1 piece).  The wait in front of a step's barrier leaves the
    // pieces of the PREVIOUS step in flight: 2, or 3 when that step carried a patch piece (the patch piece is issued BEFORE the tile pieces
    // of its step, so "everything but the last step's pieces" includes the tile this step reads).  The barrier of step s also says that
    // every wave's fragment reads of step s - 1 have returned (they fed MFMAs that were issued in front of it): the requests behind it may
    // overwrite slot (s + 2) % 3 = (s - 1) % 3.
#pragma unroll
    for (int i = 0; i < 4; ++i) patch_issue(i, 0);
    b_issue(0, 0, 0);
    b_issue(0, 1, 1);
    CHD_STAMP(CHD_TR_STEPS * 3 + 1);
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            constexpr int READS = 2 * T::TM + (DGRAD ? 2 : 4) * T::TN, MF = 2 * T::TM * T::TN;
            CHD_STAMP_STEP(9 * c + tp, 0);
            if (tp >= 1 && tp <= 4) ring_wait_vmcnt<3>();
            else ring_wait_vmcnt<2>();
            CHD_STAMP_STEP(9 * c + tp, 1);
            if constexpr ((DETR_ABLATE & 8) == 0) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            CHD_STAMP_STEP(9 * c + tp, 2);
            auto issue = [&]() {
                if constexpr ((DETR_ABLATE & 2) == 0) {
                    int cc = c;
                    asm volatile("" : "+s"(cc));         // opaque: the descriptors of the nine steps are computed where they are used, not kept in SGPRs
                    if (tp < 4) patch_issue(tp, cc + 1);
                    b_issue(cc + (tp + 2) / 9, (tp + 2) % 9, (tp + 2) % NB);
                }
            };
            if constexpr (!LATE) {
                issue();
                read_tap(c & 1, tp % NB, tp);
                mma_half(0);
                mma_half(1);
                if constexpr ((DETR_ABLATE & 17) == 0) {
                    sgb_ds_read<READS>();
                    sgb_mfma_block<MF, 0, 0, 0>();
                }
            } else {
                read_tap(c & 1, tp % NB, tp);
                mma_half(0);
                if constexpr ((DETR_ABLATE & 17) == 0) {
                    sgb_ds_read<READS>();
                    sgb_mfma_block<MF / 2, 0, 0, 0>();
                }
                __builtin_amdgcn_sched_barrier(0);
                issue();
                __builtin_amdgcn_sched_barrier(0);
                mma_half(1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    CHD_STAMP(CHD_TR_STEPS * 3 + 2);
    ring_wait_vmcnt<0>();                                // the trailing empty requests still write (zeros) into the ring the epilogue reuses
    __syncthreads();
    halo_epilogue<BN, TH, false>(acc, reinterpret_cast<float *>(chd_smem), reinterpret_cast<unsigned short *>(a.dst), a, n, h0, w0, n0, wm, wn, lane, wave);
#if (DETR_ABLATE & 64) != 0
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    CHD_STAMP(CHD_TR_STEPS * 3 + 3);
    if (tr_w && lane == 0) trl[CHD_TR_STEPS * 3 + 5] = (long long)__builtin_amdgcn_s_memrealtime();
    if (tr_w) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = lane; i < CHD_TR_N; i += 64) chd_trace[tr_sel][i] = trl[i];
    }
#endif
#endif
}

// DETR_HIP_CONV_DMA: 2 = off (register-staged kernel), 3 = requests in front of the fragment reads (A/B of their placement)
static int launch_conv_halo_dma(const ConvArgs &a0, bool dgrad, hipStream_t s) {
    ConvArgs a = a0;
    a.Hp = cdiv(a.Hd, CHD_TH);
    a.Wp = cdiv(a.Wd, CH_TW);
    a.tiles_m = a.N * a.Hp * a.Wp;
    a.tiles_n = cdiv(a.Cd, CHD_BN);
    const bool late = tune(T_CONV_DMA) != 3;
    const int smem = CHD_SMEM + ((DETR_ABLATE & 64) != 0 ? 1024 : 0);
    const int k = (dgrad ? 1 : 0) + (late ? 2 : 0);
    const void *fns[4] = {reinterpret_cast<const void *>(conv3x3_halo_dma_bf16_kernel<false, false>), reinterpret_cast<const void *>(conv3x3_halo_dma_bf16_kernel<true, false>),
                          reinterpret_cast<const void *>(conv3x3_halo_dma_bf16_kernel<false, true>), reinterpret_cast<const void *>(conv3x3_halo_dma_bf16_kernel<true, true>)};
    static bool reserved[4] = {false, false, false, false};
    if (!reserved[k]) {
        hipError_t err = hipFuncSetAttribute(fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        DETR_REQUIRE(err == hipSuccess, "conv3x3 (halo, DMA): cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(err));
        reserved[k] = true;
    }
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(GEMM_THREADS);
    switch (k) {
        case 0: hipLaunchKernelGGL((conv3x3_halo_dma_bf16_kernel<false, false>), grid, block, smem, s, a); break;
        case 1: hipLaunchKernelGGL((conv3x3_halo_dma_bf16_kernel<true, false>), grid, block, smem, s, a); break;
        case 2: hipLaunchKernelGGL((conv3x3_halo_dma_bf16_kernel<false, true>), grid, block, smem, s, a); break;
        default: hipLaunchKernelGGL((conv3x3_halo_dma_bf16_kernel<true, true>), grid, block, smem, s, a); break;
    }
    return 0;
}

}  // namespace detr

#if (DETR_ABLATE & 64) != 0
extern "C" int detr_hip_debug_conv_trace(long long *out) {       // experiment builds only: 3 x CHD_TR_N stamps of the last launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(detr::chd_trace), sizeof(detr::chd_trace)) == hipSuccess ? 0 : -1;
}
#endif
