// gemm_x3pp.h -- EXPERIMENT, not part of the library (round 6; profiles/r06_ab_results.txt #7): the f32x3 128 x 128 GEMM as one 8-wave
// workgroup in "ping-pong" form.  To build it: include this file behind gemm_x3.h in csrc/gemm_f32.hip and launch gemm_x3pp_kernel from
// launch_x3<128, 128> with grid.x = ceil(tiles / 2), 512 threads (the dispatch that was measured is in the git history: commit "f32x3 ping-pong").
#pragma once
#include "gemm_x3.h"
namespace detr {
// ---- ping-pong form: ONE workgroup of 8 waves = two groups of 4, each with its own output tile, LDS images and register set, held in
// ANTI-PHASE by the workgroup barrier: while group 0 splits and stores its K tile t (VALU + LDS writes), group 1 multiplies its tile t - 1
// (matrix pipe), then they swap.  Why: with two independent 4-wave workgroups per CU (gemm_x3_kernel) the pair starts together and stays
// IN phase -- both split, then both multiply --, and the launch time is the SUM of the two phases: the timing-experiment builds
// (scripts/experiments/x3_ablate.sh; M33600 N256 K1024) read 155 us whole, 87 us with one MFMA term of six, 130 us without the split
// arithmetic.  A SIMD overlaps one wave's MFMAs with another wave's VALU only when they are in different phases (MI355X_MICROARCH.md,
// "Two waves per SIMD"); the barrier makes that the only possible state.  Same arithmetic in the same order per accumulator: bit-identical
// to gemm_x3_kernel.  No fused row sums in this form (the host keeps gemm_x3_kernel for those launches).
template <int BM, int BN, bool AK, bool BKC>
__global__ __launch_bounds__(2 * GEMM_THREADS, 1) void gemm_x3pp_kernel(GemmArgs g) {
    constexpr int WGM = 2, WGN = 2;
    using T = TileCfg<BM, BN, WGM, WGN>;
    constexpr int GRP_BYTES = X3SmemBytes<BM, BN, 1>::VALUE;
    __shared__ __attribute__((aligned(16))) char smem_all[2 * GRP_BYTES];
    const int grp = threadIdx.x >> 8;                       // wave-uniform
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    char *smem_raw = smem_all + grp * GRP_BYTES;
    X3Smem<BM, BN, 1> &sm = *reinterpret_cast<X3Smem<BM, BN, 1> *>(smem_raw);
    const int wm = wave / WGN, wn = wave % WGN;
    int pair, zidx;
    gemm_work_item(g, pair, zidx);                          // (the grid counts tile PAIRS)
    const int tiles = g.tiles_m * g.tiles_n;
    int id = 2 * pair + grp;
    const bool valid = id < tiles;                          // odd tile count: the last pair's second group multiplies a copy and stores nothing
    id = valid ? id : tiles - 1;
    const int tn = id % g.tiles_n, tm = id / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = zidx % g.split_k;
    const int zb = zidx / g.split_k;
    const int z0 = zb / g.batch_inner, z1 = zb % g.batch_inner;
    const float *A = g.A + z0 * g.sA0 + z1 * g.sA1;
    const float *B = g.B + z0 * g.sB0 + z1 * g.sB1;
    float *C = g.C + z0 * g.sC0 + z1 * g.sC1 + (long long)split * g.part_stride;
    const int nkt = (g.K + X3_BK - 1) / X3_BK;
    const int per = (nkt + g.split_k - 1) / g.split_k;
    const int kt0 = split * per;
    const int kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;                                 // (workgroup-uniform)
    const int kend = min(g.K, kt1 * X3_BK);

    using LA = typename std::conditional<AK, X3LoaderK<BM>, X3LoaderMN<BM>>::type;
    using LB = typename std::conditional<BKC, X3LoaderK<BN>, X3LoaderMN<BN>>::type;
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
    constexpr int NA = BM / 32, NB_ = BN / 32;
    float4 ra[NA], rb[NB_];
    auto produce = [&](const int kt) {       // split + store tile kt from the registers, request tile kt + 1
        if constexpr (AK) x3_store_k<BM>(sm.A[0], ra, tid);
        else x3_store_mn<BM>(sm.A[0], ra, tid);
        if constexpr (BKC) x3_store_k<BN>(sm.B[0], rb, tid);
        else x3_store_mn<BN>(sm.B[0], rb, tid);
        la.load((kt + 1) * X3_BK, kend, ra);
        lb.load((kt + 1) * X3_BK, kend, rb);
    };
    auto consume = [&]() {
#pragma unroll
        for (int ks = 0; ks < X3_BK; ks += 16) {
            Split3Frag a[T::TM], b[T::TN];
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi) a[mi] = x3_frag<BM, AK>(sm.A[0], wm * T::WTM + mi * 32, ks, lane);
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) b[ni] = x3_frag<BN, BKC>(sm.B[0], wn * T::WTN + ni * 32, ks, lane);
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni) acc[mi][ni] = split3_mma<DETR_SPLIT3_TERMS>(a[mi], b[ni], acc[mi][ni]);
        }
    };
    la.load(kt0 * X3_BK, kend, ra);
    lb.load(kt0 * X3_BK, kend, rb);
    for (int kt = kt0; kt < kt1; ++kt) {
        if (grp == 0) produce(kt);
        else if (kt > kt0) consume();        // group 1's tile kt - 1
        lds_barrier();
        if (grp == 0) consume();
        else produce(kt);
        lds_barrier();
    }
    if (grp == 1) consume();
    __syncthreads();
    if (g.slab_ts) {
        if (valid) store_slab_ts<BM, BN, WGM, WGN>(acc, C + (long long)id * (BM * BN), wave, lane);
        return;
    }
    epilogue<BM, BN, WGM, WGN, false>(acc, reinterpret_cast<float *>(smem_raw), C, g.ldc, valid ? g.M : 0, g.N, m0, n0, wm, wn, lane, wave, g.e, false);
}

}  // namespace detr
