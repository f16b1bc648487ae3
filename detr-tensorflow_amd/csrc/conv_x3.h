// conv_x3.h -- 3x3 convolution (forward, input gradient, weight gradient) in the "f32x3" mode (detr_conv3x3_desc.compute = 2): fp32 tensors,
// fp32 accuracy, bf16 matrix pipe -- the SECOND form of gemm_x3.h applied to the implicit GEMM of conv_f32.hip:
//   * every fp32 value is split into its three bf16 pieces ONCE, on the way from the request registers into LDS (split3_pair), not by every
//     wave that reads it (the first form, conv3x3_kernel<.., SPLIT3 = true>: each value was split by two waves and gathered with ds_read_b32);
//   * LDS holds three bf16 images per operand in the bf16 engine's layouts: [row][k] rows of 80 B read with ds_read_b128 for a channel-
//     contiguous operand, the transpose-read image + ds_read_b64_tr_b16 for a row-contiguous one; 32-deep K tiles (32 channels of one tap,
//     or 32 output pixels of the weight gradient);
//   * one LDS buffer (60 KB at 128 x 128: two workgroups per CU cover each other's split phase), 6 MFMAs per fragment pair (split3_mma).
// Reference semantics: detr_tf/networks/resnet_backbone.py:116-137 (the 3x3 convolution of a bottleneck) and its tape.gradient.
#pragma once
#include "gemm_x3.h"

namespace detr {

// A operand of forward / input gradient: rows = destination pixels, 32 source channels of one tap per K tile.
// X3LoaderK's thread map: rows (t >> 3) + 32 i, channel offset (t & 7) * 4 (x3_store_k relies on it)
template <int BM, bool DGRAD>
struct X3LoaderConvA {
    static constexpr int NV = BM / 32;
    BufSrc src;
    int n_[NV], h_[NV], w_[NV];
    bool ok[NV];
    int kq;
    __device__ __forceinline__ void init(const ConvArgs &a, int m0, int tid) {
        src.init(a.src, (long long)a.N * a.Hs * a.Ws * a.Cs);
        kq = (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int m = m0 + (tid >> 3) + 32 * i;
            ok[i] = m < a.M;
            const int mm = ok[i] ? m : 0;
            int wd, hd;
            if (DGRAD && a.par_on) {
                const int t = mm / a.Wp;
                wd = 2 * (mm - t * a.Wp) + a.pw;
                hd = 2 * (t % a.Hp) + a.ph;
                n_[i] = t / a.Hp;
            } else {
                wd = mm % a.Wd;
                const int t = mm / a.Wd;
                hd = t % a.Hd;
                n_[i] = t / a.Hd;
            }
            h_[i] = DGRAD ? hd + a.pad : hd * a.stride - a.pad;
            w_[i] = DGRAD ? wd + a.pad : wd * a.stride - a.pad;
        }
    }
    // halo / stride-parity / tile-edge lanes take the out-of-range offset: the descriptor returns zeros, no branch
    __device__ __forceinline__ void load_tap(const ConvArgs &a, int kh, int kw, int kc, float4 (&r)[NV]) const {
        const int c0 = kc * X3_BK + kq;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int hs, ws;
            bool v = ok[i];
            if (DGRAD) {
                const int th = h_[i] - kh, tw = w_[i] - kw;
                v = v && th >= 0 && tw >= 0;
                if (a.stride == 2) {
                    v = v && ((th & 1) == 0) && ((tw & 1) == 0);
                    hs = th >> 1;
                    ws = tw >> 1;
                } else {
                    hs = th;
                    ws = tw;
                }
                v = v && hs < a.Hs && ws < a.Ws;
            } else {
                hs = h_[i] + kh;
                ws = w_[i] + kw;
                v = v && hs >= 0 && ws >= 0 && hs < a.Hs && ws < a.Ws;
            }
            const unsigned off = ((unsigned)((n_[i] * a.Hs + hs) * a.Ws + ws) * (unsigned)a.Cs + (unsigned)c0) * 4u;
            r[i] = src.ld4(v ? off : BUF_OOB);
        }
    }
};

template <int BM, int BN, bool DGRAD>
__global__ __launch_bounds__(GEMM_THREADS, (BM * BN >= 128 * 128) ? 2 : (BM * BN >= 128 * 64 ? 3 : 4)) void conv3x3_x3_kernel(ConvArgs a) {
    constexpr int WGM = 2, WGN = 2;
    using T = TileCfg<BM, BN, WGM, WGN>;
    __shared__ __attribute__((aligned(16))) char smem_raw[X3SmemBytes<BM, BN, 1>::VALUE];
    X3Smem<BM, BN, 1> &sm = *reinterpret_cast<X3Smem<BM, BN, 1> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = id % a.tiles_n, tm = id / a.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int cpt = a.Cs / X3_BK;    // K tiles per tap (the host sends Cs % 32 == 0 here)
    const int nkt = (a.par_on ? a.nth * a.ntw : 9) * cpt;
    const long long tapstride = (long long)a.Ci * a.Co;

    X3LoaderConvA<BM, DGRAD> la;
    la.init(a, m0, tid);
    // B operand per tap: fwd  B[k = ci][n = co] = w[tap][ci][co]  (n contiguous, ld = Co, K = Ci)
    //                    dgrad B[k = co][n = ci] = w[tap][ci][co]  (k contiguous, ld = Co, K = Co)
    using LB = typename std::conditional<DGRAD, X3LoaderK<BN>, X3LoaderMN<BN>>::type;
    LB lb;
    lb.init(a.w, a.Co, n0, a.Cd, a.Cs, true, tid);

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    constexpr int NA = BM / 32, NB_ = BN / 32;
    float4 ra[NA], rb[NB_];
    // K tiles are requested strictly in order: tap and channel tile are running counters
    const int tap_cols = a.par_on ? a.ntw : 3;
    int it_kc = 0, it_ti = 0, it_tj = 0;
    auto load_ab = [&]() {
        const int kh = a.par_on ? a.kh0 + 2 * it_ti : it_ti;
        const int kw = a.par_on ? a.kw0 + 2 * it_tj : it_tj;
        la.load_tap(a, kh, kw, it_kc, ra);
        lb.base = a.w + (long long)(kh * 3 + kw) * tapstride;
        lb.load(it_kc * X3_BK, a.Cs, rb);
        if (++it_kc == cpt) {
            it_kc = 0;
            if (++it_tj == tap_cols) {
                it_tj = 0;
                ++it_ti;
            }
        }
    };
    load_ab();
    for (int kt = 0; kt < nkt; ++kt) {
        x3_store_k<BM>(sm.A[0], ra, tid);
        if constexpr (DGRAD) x3_store_k<BN>(sm.B[0], rb, tid);
        else x3_store_mn<BN>(sm.B[0], rb, tid);
        if (kt + 1 < nkt) load_ab();          // (wave-uniform) the next tile's requests fly under this tile's MFMA phase
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < X3_BK; ks += 16) {
            Split3Frag fa[T::TM], fb[T::TN];
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi) fa[mi] = x3_frag<BM, true>(sm.A[0], wm * T::WTM + mi * 32, ks, lane);
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) fb[ni] = x3_frag<BN, DGRAD>(sm.B[0], wn * T::WTN + ni * 32, ks, lane);
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni) acc[mi][ni] = split3_mma<DETR_SPLIT3_TERMS>(fa[mi], fb[ni], acc[mi][ni]);
        }
        lds_barrier();
    }
    __syncthreads();
    epilogue<BM, BN, WGM, WGN>(acc, reinterpret_cast<float *>(smem_raw), a.dst, a.Cd, a.M, a.Cd, m0, n0, wm, wn, lane, wave, a.e);
}

// A operand of the weight gradient: rows of the reduction = output pixels, columns = input channels ci0 .. ci0 + BM of the tap's source pixel.
// X3LoaderMN's unit map (u = t + 256 i: column group 16 * ((u >> 4) & (NB - 1)) + 4 * (u & 3), reduction row 4 * (u / (16 NB)) + ((u >> 2) & 3)):
// x3_store_mn writes the transpose-read image linearly in that order
template <int BM>
struct X3LoaderWgradA {
    static constexpr int NB = BM / 16;
    static constexpr int NV = BM / 32;
    BufSrc src;
    int n_[NV], h_[NV], w_[NV], m_[NV];
    int col[NV];
    int kh, kw;
    __device__ __forceinline__ void init(const ConvWgradArgs &a, int ci0, int tap, int m_begin, int tid) {
        src.init(a.x, (long long)a.N * a.Hi * a.Wi * a.Ci);
        kh = tap / 3;
        kw = tap - kh * 3;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + 256 * i;
            const int kr = 4 * (u / (16 * NB)) + ((u >> 2) & 3);
            col[i] = ci0 + 16 * ((u >> 4) & (NB - 1)) + 4 * (u & 3);
            const int m = m_begin + kr;
            m_[i] = m;
            const int wo = m % a.Wo;
            const int t = m / a.Wo;
            w_[i] = wo;
            h_[i] = t % a.Ho;
            n_[i] = t / a.Ho;
        }
    }
    __device__ __forceinline__ void load(const ConvWgradArgs &a, int m_end, float4 (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int hs = h_[i] * a.stride - a.pad + kh;
            const int ws = w_[i] * a.stride - a.pad + kw;
            const bool v = (m_[i] < m_end) && hs >= 0 && ws >= 0 && hs < a.Hi && ws < a.Wi && (col[i] < a.Ci);
            const unsigned off = ((unsigned)((n_[i] * a.Hi + hs) * a.Wi + ws) * (unsigned)a.Ci + (unsigned)col[i]) * 4u;
            r[i] = src.ld4(v ? off : BUF_OOB);
        }
    }
    __device__ __forceinline__ void advance(const ConvWgradArgs &a) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            m_[i] += X3_BK;
            w_[i] += X3_BK;
            while (w_[i] >= a.Wo) {
                w_[i] -= a.Wo;
                h_[i] += 1;
            }
            while (h_[i] >= a.Ho) {
                h_[i] -= a.Ho;
                n_[i] += 1;
            }
        }
    }
};

// grid = (ci tiles * co tiles, 9 taps, row splits) as conv3x3_wgrad_kernel (any rows_per_split: the last 32-row tile of a split is cut by m_end)
template <int BM, int BN>
__global__ __launch_bounds__(GEMM_THREADS, (BM * BN >= 128 * 128) ? 2 : 4) void conv3x3_wgrad_x3_kernel(ConvWgradArgs a) {
    constexpr int WGM = 2, WGN = 2;
    using T = TileCfg<BM, BN, WGM, WGN>;
    __shared__ __attribute__((aligned(16))) char smem_raw[X3SmemBytes<BM, BN, 1>::VALUE];
    X3Smem<BM, BN, 1> &sm = *reinterpret_cast<X3Smem<BM, BN, 1> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tn = blockIdx.x % a.tiles_n, tm = blockIdx.x / a.tiles_n;
    const int ci0 = tm * BM, co0 = tn * BN;
    const int tap = blockIdx.y;
    const int m_begin = blockIdx.z * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    if (m_begin >= m_end) return;
    const int nkt = (m_end - m_begin + X3_BK - 1) / X3_BK;

    X3LoaderWgradA<BM> la;
    la.init(a, ci0, tap, m_begin, tid);
    X3LoaderMN<BN> lb;
    lb.init(a.dy, a.Co, co0, a.Co, a.M, true, tid);

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    constexpr int NA = BM / 32, NB_ = BN / 32;
    float4 ra[NA], rb[NB_];
    la.load(a, m_end, ra);
    lb.load(m_begin, m_end, rb);
    for (int kt = 0; kt < nkt; ++kt) {
        x3_store_mn<BM>(sm.A[0], ra, tid);
        x3_store_mn<BN>(sm.B[0], rb, tid);
        if (kt + 1 < nkt) {
            la.advance(a);
            la.load(a, m_end, ra);
            lb.load(m_begin + (kt + 1) * X3_BK, m_end, rb);
        }
        lds_barrier();
#pragma unroll
        for (int ks = 0; ks < X3_BK; ks += 16) {
            Split3Frag fa[T::TM], fb[T::TN];
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi) fa[mi] = x3_frag<BM, false>(sm.A[0], wm * T::WTM + mi * 32, ks, lane);
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) fb[ni] = x3_frag<BN, false>(sm.B[0], wn * T::WTN + ni * 32, ks, lane);
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni) acc[mi][ni] = split3_mma<DETR_SPLIT3_TERMS>(fa[mi], fb[ni], acc[mi][ni]);
        }
        lds_barrier();
    }
    __syncthreads();
    if (a.slab_ts) {
        float *slab = a.dw + (long long)blockIdx.z * a.part_stride + ((long long)(tap * a.tiles_m + tm) * a.tiles_n + tn) * (BM * BN);
        store_slab_ts<BM, BN, WGM, WGN>(acc, slab, wave, lane);
        return;
    }
    float *dw = a.dw + (long long)tap * a.Ci * a.Co + (long long)blockIdx.z * a.part_stride;
    epilogue<BM, BN, WGM, WGN, false>(acc, reinterpret_cast<float *>(smem_raw), dw, a.Co, a.Ci, a.Co, ci0, co0, wm, wn, lane, wave, a.e);
}

}  // namespace detr
