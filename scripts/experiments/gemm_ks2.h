// gemm_ks2.h -- EXPERIMENT, not part of the build (round 3: correct, but not faster -- M256 N1024 K33600 sk16 42.1 us with it, 42.3
// without; cold 60.3 vs 57.7; the split-K weight gradients are not short of waves).
// Split-K weight-gradient GEMM with the K range of a workgroup cut over TWO wave groups (included by gemm_f32.hip).
//
// The long-K weight gradients (C[M, N] += A[K, M]^T B[K, N], K = B*H*W = 33 600 .. 534 400) run as 128x128 tiles x split_k
// launches of ~256 workgroups: 512 workgroups measured slower because every split writes (and the reduction re-reads) a full
// M x N fp32 slab.  But 256 workgroups of four waves are ONE wave per SIMD: the K loop is a chain of barrier-separated round
// trips with nothing to overlap them with.  Here a workgroup has eight waves -- two groups of four, each with its own LDS
// buffers and operand pipeline over half of the workgroup's K range -- i.e. the parallelism of twice the splits without their
// slabs (the same idea as the wave-pair split of the attention kernels).  The second group's accumulators are added to the
// first's through LDS in a fixed order (deterministic), and the first group runs the unchanged epilogue.
// Arithmetic: the two halves are summed as (first half) + (second half) instead of one running sum, so results differ from the
// one-group kernel in the last bits of the fp32 partial sums (tests/test_gpu_kernels.py::test_gemm_bf16_two_group_split_k).
#pragma once
#include "gemm_core.h"
#include "gemm_bf16_core.h"

namespace detr {

template <int BM, int BN, bool AK, bool BKC, int BK>
__global__ __launch_bounds__(2 * GEMM_THREADS, 1) void gemm_bf16c_ks2_kernel(GemmArgs g) {
    using T = TileCfg<BM, BN, 2, 2>;
    using Smem = BfSmem<BM, BN, BK>;
    constexpr int GROUP_BYTES = (int)sizeof(Smem);
    constexpr int MERGE_BYTES = GEMM_THREADS * T::TM * T::TN * 16 * 4;       // one group's accumulators
    constexpr int STAGE_BYTES = StageCfg<BN, 2>::BYTES;
    constexpr int SMEM = 2 * GROUP_BYTES > MERGE_BYTES ? (2 * GROUP_BYTES > STAGE_BYTES ? 2 * GROUP_BYTES : STAGE_BYTES)
                                                       : (MERGE_BYTES > STAGE_BYTES ? MERGE_BYTES : STAGE_BYTES);
    __shared__ __attribute__((aligned(16))) char smem_raw[SMEM];
    const int grp = threadIdx.x >> 8;                                        // K half of this wave group
    const int tid = threadIdx.x & (GEMM_THREADS - 1), lane = tid & 63, wave = tid >> 6;
    Smem &sm = *reinterpret_cast<Smem *>(smem_raw + grp * GROUP_BYTES);
    const int wm = wave >> 1, wn = wave & 1;
    int id, z;
    gemm_work_item(g, id, z);
    const int tn = id % g.tiles_n, tm = id / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = z % g.split_k;                                         // (no batch: the host dispatch)
    float *C = g.C + (long long)split * g.part_stride;
    const int nkt = (g.K + BF_BK - 1) / BF_BK;
    const int per = (nkt + g.split_k - 1) / g.split_k;
    const int kt0_32 = split * per;
    const int kt1_32 = min(nkt, kt0_32 + per);
    if (kt0_32 >= kt1_32) return;
    const int kbeg_wg = kt0_32 * BF_BK;
    const int kend_wg = min(g.K, kt1_32 * BF_BK);
    // both groups run the same number of iterations (they share the barrier); a shorter second half ends in tiles whose requests
    // all resolve to the out-of-range offset (zeros, no traffic)
    const int kt1 = ((kend_wg - kbeg_wg + BK - 1) / BK + 1) / 2;
    const int kbeg = kbeg_wg + grp * kt1 * BK;
    const int kend = min(kend_wg, kbeg + kt1 * BK);

    using LA = typename std::conditional<AK, LoaderKh<BM, BK>, LoaderMNth<BM, true, BK>>::type;
    using LB = typename std::conditional<BKC, LoaderKh<BN, BK>, LoaderMNth<BN, true, BK>>::type;
    constexpr int NRA = LA::NREG, NRB = LB::NREG;
    LA la;
    LB lb;
    la.init(g.A, g.lda, m0, g.M, g.K, true, tid);
    lb.init(g.B, g.ldb, n0, g.N, g.K, true, tid);
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // operand pipeline of gemm_bf16c_body (two K tiles deep, unconditional requests, LDS-only barrier)
    typename LA::Reg ra0[NRA], ra1[NRA];
    typename LB::Reg rb0[NRB], rb1[NRB];
    la.load(kbeg, kend, ra0);
    lb.load(kbeg, kend, rb0);
    la.store(sm.A[0], ra0);
    lb.store(sm.B[0], rb0);
    la.load(kbeg + BK, kend, ra0);
    lb.load(kbeg + BK, kend, rb0);
    la.load(kbeg + 2 * BK, kend, ra1);
    lb.load(kbeg + 2 * BK, kend, rb1);
    lds_barrier();
    auto iter = [&](const int kt, const int cur, typename LA::Reg (&rpa)[NRA], typename LB::Reg (&rpb)[NRB]) {
        la.store(sm.A[cur ^ 1], rpa);
        lb.store(sm.B[cur ^ 1], rpb);
        la.load(kbeg + (kt + 3) * BK, kend, rpa);
        lb.load(kbeg + (kt + 3) * BK, kend, rpb);
        mma_ktile_bf16<BM, BN, 2, 2, !AK, !BKC, BK>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
        lds_barrier();
    };
    {
        int kt = 0;
        for (; kt + 2 <= kt1; kt += 2) {
            iter(kt, 0, ra0, rb0);
            iter(kt + 1, 1, ra1, rb1);
        }
        if (kt < kt1) iter(kt, 0, ra0, rb0);
    }
    // merge: second group -> LDS ([register][thread]: conflict-free) -> first group adds, then runs the epilogue alone
    __syncthreads();
    float *mg = reinterpret_cast<float *>(smem_raw);
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mg[((i * T::TN + j) * 16 + r) * GEMM_THREADS + tid] = acc[i][j][r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += mg[((i * T::TN + j) * 16 + r) * GEMM_THREADS + tid];
    }
    __syncthreads();                 // the staging region of the epilogue aliases the merge buffer
    // (the second group stays resident up to here so that every barrier above is reached by all eight waves)
    if (grp == 0)
        epilogue<BM, BN, 2, 2, false>(acc, reinterpret_cast<float *>(smem_raw), C, g.ldc, g.M, g.N, m0, n0, wm, wn, lane, wave, g.e, false);
}

template <int BM, int BN, int BK>
static void launch_gemm_ks2(const GemmArgs &g, bool ak, bool bk, hipStream_t s) {
    GemmArgs a = g;
    a.tiles_m = cdiv(g.M, BM);
    a.tiles_n = cdiv(g.N, BN);
    const dim3 grid((unsigned)(a.tiles_m * a.tiles_n), 1, (unsigned)g.split_k), block(2 * GEMM_THREADS);
    if (ak && bk) hipLaunchKernelGGL((gemm_bf16c_ks2_kernel<BM, BN, true, true, BK>), grid, block, 0, s, a);
    else if (ak) hipLaunchKernelGGL((gemm_bf16c_ks2_kernel<BM, BN, true, false, BK>), grid, block, 0, s, a);
    else if (bk) hipLaunchKernelGGL((gemm_bf16c_ks2_kernel<BM, BN, false, true, BK>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_bf16c_ks2_kernel<BM, BN, false, false, BK>), grid, block, 0, s, a);
}

}  // namespace detr
