// gemm_dira.h -- EXPERIMENT, not part of the build (round 3; measured slower than the LDS engine, see
// profiles/r03_micro_gemm_direct_a_experiment.txt; it needs frag_tr_at, a frag_tr with a caller-chosen k, in gemm_bf16_core.h).
// "direct-A" bf16 GEMM: the tile engine's variant for C[M, N] = A[M, K] * B with A bf16 and K-contiguous
// (every forward / input-gradient GEMM of the step: the 1x1 convolutions on bf16 activations, the transformer's linears).
//
// Why: the 32-deep LDS pipeline of gemm_bf16c_body pays, per 32 k of a 128x128 tile, one workgroup barrier, 16 KB of
// ds_write_b128 (A and B), 32 KB of fragment reads and a vmcnt wait for 8 MFMAs per wave -- measured 3000 cycles per CU for three
// such iterations (M33600 N1024 K512: 73 us, 14 % of the MFMA peak, 3.5x the HBM bound).  The MFMA A operand of a lane is 8
// consecutive k of ONE row of A, i.e. 16 contiguous bytes of global memory when A is K-contiguous: it needs no LDS at all.
//
//   * wave tile 32 rows x 128 columns (TM = 1, TN = 4), workgroup = 4 waves stacked in M (128 x 128): every A row is loaded
//     by exactly one wave, straight into the registers the MFMA reads (one 16-byte request per lane and k-step), two
//     128-deep chunks ahead (2 x 8 k-steps x 4 VGPRs = 64 VGPRs);
//   * only B goes through LDS: 128-deep chunks (128 columns x 128 k = 32 KB of bf16), double buffered, register staged one
//     chunk ahead with the tile engine's own loaders (LoaderKh / LoaderMNth at BK = 128: same LDS images, same fragment reads);
//   * one barrier per 128 k (32 MFMAs per wave) instead of one per 32 k (8 MFMAs), no ds_write for A, B fragment reads only:
//     per 128 k and 128x128 outputs 32 KB of LDS writes + 64 KB of reads instead of 64 KB + 128 KB;
//   * L1 / texture path: one 1 KB request per wave and 4 MFMAs (128 MFMA cycles) -> 32 B/clk/CU at full MFMA rate, half the
//     L1's 64 B/clk; LDS: 4 KB of fragment reads per 4 MFMAs and wave -> 128 B/clk/CU at full rate, half its peak;
//   * same MFMA instruction, operand roles and epilogue<> as gemm_bf16c_body; the k ORDER inside a 64-deep group differs (see
//     k_of below), so results agree with the 32-deep variant to fp32 summation order, not bit for bit
//     (tests/test_gpu_kernels.py::test_gemm_bf16_direct_a_matches_the_staged_engine).
// LDS: 2 x 128 x (128 + 8) x 2 B = 68 KB (the epilogue's 66 KB of wave-private staging aliases it): two workgroups per CU.
#pragma once
#include "gemm_core.h"
#include "gemm_bf16_core.h"

namespace detr {

constexpr int DA_BM = 128, DA_BN = 128, DA_BK = 128;

struct DiraSmem {
    unsigned short B[2][DA_BN][DA_BK + 8];
};

template <bool BKC>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16c_dira_kernel(GemmArgs g) {
    using T = TileCfg<DA_BM, DA_BN, 4, 1>;
    constexpr int KS = DA_BK / 16;                                   // k-steps (MFMA depth 16) per chunk
    constexpr int SMEM = (int)sizeof(DiraSmem) > StageCfg<DA_BN, 1>::BYTES ? (int)sizeof(DiraSmem) : StageCfg<DA_BN, 1>::BYTES;
    __shared__ __attribute__((aligned(16))) char smem_raw[SMEM];
    DiraSmem &sm = *reinterpret_cast<DiraSmem *>(smem_raw);
    int tile, z;
    gemm_work_item(g, tile, z);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tn = tile % g.tiles_n, tm = tile / g.tiles_n;
    const int m0 = tm * DA_BM, n0 = tn * DA_BN;
    const int K = g.K;

    // A: lane l of wave w supplies row m0 + 32 w + (l & 31), k = 8 (l >> 5) .. +7 of every k-step
    BufSrc asrc;
    asrc.init_bytes(g.A, ((long long)(g.M - 1) * g.lda + K) * 2);
    // k order inside a 64-deep group: k-step j (0..3) of the group multiplies k = 32 h + 8 j .. + 7 for lane half h = l >> 5 (the
    // MFMA only needs A and B fragments to agree on which k a (half, element) slot carries).  A lane's four requests of a group are
    // then 64 CONTIGUOUS bytes of its row, issued back to back: its 128-byte line is fetched from L2 once.  With the natural order
    // (k = 16 j + 8 h) a line is touched by four requests a k-step apart, and with 0.5 MB of lines in flight per CU the 32 KB L1
    // has dropped it in between: 4x the L2 -> L1 line traffic (measured: 62 us against 46 us of the LDS engine on M33600 N256 K1024)
    const int kl = (lane >> 5) * 32;
    const int row = m0 + wave * 32 + (lane & 31);
    const unsigned arow = row < g.M ? (unsigned)((long long)row * g.lda * 2) : BUF_OOB;
    auto k_of = [&](int s) -> int { return (s >> 2) * 64 + kl + (s & 3) * 8; };      // chunk-relative first k of this lane in k-step s
    auto a_frag = [&](int kc, int s) -> uint4 {          // (requests past K or M resolve to the out-of-range offset: zeros, no traffic)
        const int k = kc + k_of(s);
#if defined(DETR_DIRA_ABLATE) && (DETR_DIRA_ABLATE & 1)            // timing probe: no A requests
        return make_uint4(k, s, 0x3f803f80u, 0x3f803f80u);
#endif
        return asrc.ld16((arow != BUF_OOB && k + 8 <= K) ? arow + 2u * (unsigned)k : BUF_OOB);
    };
    using LB = typename std::conditional<BKC, LoaderKh<DA_BN, DA_BK>, LoaderMNth<DA_BN, true, DA_BK>>::type;
    LB lb;
    lb.init(g.B, g.ldb, n0, g.N, K, false, tid);
    typename LB::Reg rb[LB::NREG];

    f32x16 acc[1][T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;

    uint4 a0[KS], a1[KS];
#if defined(DETR_DIRA_ABLATE) && (DETR_DIRA_ABLATE & 8)            // timing probe: epilogue only
    const int nch = 0;
#else
    const int nch = (K + DA_BK - 1) / DA_BK;
#endif
    lb.load(0, K, rb);
#pragma unroll
    for (int s = 0; s < KS; ++s) a0[s] = a_frag(0, s);
    lb.store(sm.B[0], rb);
    lb.load(DA_BK, K, rb);
#pragma unroll
    for (int s = 0; s < KS; ++s) a1[s] = a_frag(DA_BK, s);
    lds_barrier();

    const int l31 = lane & 31;
    // chunk c: multiply out of `a` and B[cur]; each k-step's A registers are re-requested for chunk c + 2 as soon as its MFMAs
    // are issued; at the end the staged B chunk c + 1 goes to the other buffer and chunk c + 2 is requested.  Every request is
    // unconditional (see gemm_bf16c_body: exact vmcnt waits).
    auto chunk = [&](const int c, const int cur, uint4 (&a)[KS]) {
        const unsigned short(*Bs)[DA_BK + 8] = sm.B[cur];
        auto b_frags = [&](bf16x8 (&b)[T::TN], const int s) {
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) {
                if constexpr (BKC) b[ni] = *reinterpret_cast<const bf16x8 *>(&Bs[ni * 32 + l31][k_of(s)]);
                else b[ni] = frag_tr_at<DA_BN>(Bs, ni * 32, k_of(s), lane);
            }
        };
        auto step = [&](const bf16x8 (&b)[T::TN], const int s) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, a[s]);
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) {
#if defined(DETR_DIRA_ABLATE) && (DETR_DIRA_ABLATE & 4)            // timing probe: no MFMA
                ablate_keep(av); ablate_keep(b[ni]);
#else
                acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[ni], acc[0][ni], 0, 0, 0);
#endif
            }
            a[s] = a_frag((c + 2) * DA_BK, s);
        };
        // the B fragments of k-step s + 1 are read while the MFMAs of step s run (two fragment sets)
        bf16x8 b0[T::TN], b1[T::TN];
        b_frags(b0, 0);
#pragma unroll
        for (int s = 0; s < KS; s += 2) {
            b_frags(b1, s + 1);
            step(b0, s);
            if (s + 2 < KS) b_frags(b0, s + 2);
            step(b1, s + 1);
        }
        // keep the staged B chunk's LDS stores (and with them the wait for its requests) BEHIND the MFMAs of this chunk: hoisted
        // into the chunk by the scheduler, they put the round trip of requests issued 12 MFMAs earlier on the critical path
        __builtin_amdgcn_sched_barrier(0);
#if defined(DETR_DIRA_ABLATE) && (DETR_DIRA_ABLATE & 2)            // timing probe: no B requests / LDS stores (barrier kept)
        lds_barrier();
        return;
#endif
        lb.store(sm.B[cur ^ 1], rb);
        lb.load((c + 2) * DA_BK, K, rb);
        lds_barrier();
    };
    {
        int c = 0;
        for (; c + 2 <= nch; c += 2) {
            chunk(c, 0, a0);
            chunk(c + 1, 1, a1);
        }
        if (c < nch) chunk(c, 0, a0);
    }
    __syncthreads();
    float *C = g.C;
#if defined(DETR_DIRA_ABLATE) && (DETR_DIRA_ABLATE & 16)           // timing probe: no epilogue (one store per lane keeps the sums alive)
    if (acc[0][0][0] + acc[0][1][1] + acc[0][2][2] + acc[0][3][3] == 12345.0f) C[0] = 1.0f;
    return;
#endif
    epilogue<DA_BM, DA_BN, 4, 1>(acc, reinterpret_cast<float *>(smem_raw), C, g.ldc, g.M, g.N, m0, n0, wave, 0, lane, wave, g.e);
}

static inline void launch_gemm_dira(const GemmArgs &g, bool bk, hipStream_t s) {
    GemmArgs a = g;
    a.tiles_m = cdiv(g.M, DA_BM);
    a.tiles_n = cdiv(g.N, DA_BN);
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n), 1, 1);
    if (bk) hipLaunchKernelGGL((gemm_bf16c_dira_kernel<true>), grid, dim3(GEMM_THREADS), 0, s, a);
    else hipLaunchKernelGGL((gemm_bf16c_dira_kernel<false>), grid, dim3(GEMM_THREADS), 0, s, a);
}

}  // namespace detr
