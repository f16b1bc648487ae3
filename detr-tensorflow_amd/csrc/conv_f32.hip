// conv_f32.hip -- ResNet backbone convolution kernels (NHWC / HWIO, fp32) for gfx950:
//   * 3x3 implicit-GEMM convolution forward / dgrad / wgrad on the MFMA tile engine
//     (no im2col buffer: the A-operand loader gathers input pixels, zero-filling the halo);
//   * stem helpers: 3x3/s2 max pool over the zero-padded map + its backward, stride-2 subsample gather/scatter
//     (the 7x7 stem convolution itself is stem_conv.hip).
// Reference: detr_tf/networks/resnet_backbone.py:11-32,98-137 (see include/detr_hip.h).
#include "gemm_core.h"
#include "gemm_bf16_core.h"

namespace detr {

// ------------------------------------------------------------------------------------------------
// forward / dgrad : rows = destination pixels, K = 9 taps x source channels
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    int N, Hs, Ws, Cs;  // source tensor (A operand): fwd x[N,Hi,Wi,Ci]; dgrad dy[N,Ho,Wo,Co]
    int Hd, Wd, Cd;     // destination tensor: fwd y[N,Ho,Wo,Co]; dgrad dx[N,Hi,Wi,Ci]
    int stride, pad;
    int Ci, Co;
    int M;
    const float *src;
    const float *w;
    float *dst;
    int tiles_m, tiles_n;
    EpiArgs e;
    // stride-2 dgrad, one launch per destination-pixel parity class (ph, pw): rows = (n, h2, w2) with h = 2*h2 + ph,
    // w = 2*w2 + pw; only the taps with kh = kh0 (+2), kw = kw0 (+2) contribute (1, 2, 2 or 4 of the 9), so the classes
    // together do 9/4 tap-GEMMs per pixel instead of 9 with 3/4 of the rows masked.
    int par_on, Hp, Wp, ph, pw, kh0, kw0, nth, ntw;
    int w16;                     // the kernel tensor is bf16 in memory (bf16 compute only)
    int x16;                     // the source tensor (x for fwd, dy for dgrad) is bf16 in memory
    // all four parity classes in ONE launch of the halo kernel's class form (conv_halo.h): class k = 0..3 is (ph, pw) =
    // (1,1), (0,1), (1,0), (0,0) -- heaviest first --, owns workgroups [cls_off[k], cls_off[k + 1]) and cls_hp / cls_wp tiles along H / W
    int cls_off[5], cls_hp[4], cls_wp[4];
    int cls_mix;                 // 1: the classes' workgroups round-robin in the grid instead of one class after the other (conv_halo.h)
};

}  // namespace detr
#include "conv_halo.h"
#include "conv_halo_dma.h"
namespace detr {

// tap (kh, kw) of K-tile group t (t-th tap of the launch)
__device__ __forceinline__ void conv_tap(const ConvArgs &a, int t, int &kh, int &kw) {
    if (a.par_on) {
        kh = a.kh0 + 2 * (t / a.ntw);
        kw = a.kw0 + 2 * (t - (t / a.ntw) * a.ntw);
    } else {
        kh = t / 3;
        kw = t - kh * 3;
    }
}

template <int BM, bool DGRAD>
struct LoaderConvA {
    static constexpr int NV = BM / 64;
    BufSrc src;
    int n_[NV], h_[NV], w_[NV];
    bool ok[NV];
    int kq, tid;

    __device__ __forceinline__ void init(const ConvArgs &a, int m0, int tid_) {
        src.init(a.src, (long long)a.N * a.Hs * a.Ws * a.Cs);
        tid = tid_;
        kq = (tid & 3) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int m = m0 + (tid >> 2) + 64 * i;
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
    __device__ __forceinline__ void load(const ConvArgs &a, int kt, int cpt, float4 (&r)[NV]) const {
        const int tap = kt / cpt;
        int kh, kw;
        conv_tap(a, tap, kh, kw);
        load_tap(a, kh, kw, kt - tap * cpt, r);
    }
    // tap (kh, kw) and channel tile kc given by the caller (running counters in the main loop instead of divisions)
    __device__ __forceinline__ void load_tap(const ConvArgs &a, int kh, int kw, int kc, float4 (&r)[NV]) const {
        const int c0 = kc * GEMM_BK + kq;
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
    template <int LD>
    __device__ __forceinline__ void store(float (*S)[LD], const float4 (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = (tid >> 2) + 64 * i;
            S[kq + 0][row] = r[i].x;
            S[kq + 1][row] = r[i].y;
            S[kq + 2][row] = r[i].z;
            S[kq + 3][row] = r[i].w;
        }
    }
};

// SPLIT3: the K tiles run on the bf16 matrix pipe at fp32 accuracy (gemm_core.h: mma_ktile_split3; detr_conv3x3_desc.compute = 2)
template <int BM, int BN, int WGM, int WGN, bool DGRAD, bool SPLIT3 = false>
__global__ __launch_bounds__(GEMM_THREADS) void conv3x3_kernel(ConvArgs a) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    __shared__ __attribute__((aligned(16))) char smem_raw[SmemBytes<BM, BN, WGN>::VALUE];
    GemmSmem<BM, BN> &sm = *reinterpret_cast<GemmSmem<BM, BN> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = id % a.tiles_n, tm = id / a.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int cpt = a.Cs / GEMM_BK;  // K tiles per tap
    const int nkt = (a.par_on ? a.nth * a.ntw : 9) * cpt;
    const long long tapstride = (long long)a.Ci * a.Co;

    LoaderConvA<BM, DGRAD> la;
    la.init(a, m0, tid);
    // B operand per tap: fwd  B[k=ci][n=co] = w[tap][ci][co]  (n contiguous, ld = Co, K = Ci)
    //                    dgrad B[k=co][n=ci] = w[tap][ci][co]  (k contiguous, ld = Co, K = Co)
    using LB = typename std::conditional<DGRAD, LoaderK<BN>, LoaderMN<BN>>::type;
    LB lb;
    lb.init(a.w, a.Co, n0, a.Cd, a.Cs, true, tid, 9 * tapstride);   // one descriptor over the 9 taps

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[LoaderConvA<BM, DGRAD>::NV], rb[LB::NV];
    // K tiles are requested strictly in order: tap and channel tile are running counters (the kt / cpt and conv_tap()
    // divisions cost ~40 scalar instructions per 16-deep K tile)
    const int tap_cols = a.par_on ? a.ntw : 3;
    int it_kc = 0, it_ti = 0, it_tj = 0;
    auto load_ab = [&]() {
        const int kh = a.par_on ? a.kh0 + 2 * it_ti : it_ti;
        const int kw = a.par_on ? a.kw0 + 2 * it_tj : it_tj;
        la.load_tap(a, kh, kw, it_kc, ra);
        lb.load(it_kc * GEMM_BK, a.Cs, rb, (unsigned)((kh * 3 + kw) * tapstride * 4));
        if (++it_kc == cpt) {
            it_kc = 0;
            if (++it_tj == tap_cols) {
                it_tj = 0;
                ++it_ti;
            }
        }
    };
    load_ab();
    la.store(sm.A[0], ra);
    lb.store(sm.B[0], rb);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (more) load_ab();
        mma_ktile_sel<BM, BN, WGM, WGN, SPLIT3>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
        if (more) {
            la.store(sm.A[cur ^ 1], ra);
            lb.store(sm.B[cur ^ 1], rb);
        }
        __syncthreads();
        cur ^= 1;
    }
    epilogue<BM, BN, WGM, WGN>(acc, reinterpret_cast<float *>(smem_raw), a.dst, a.Cd, a.M, a.Cd, m0, n0, wm, wn, lane, wave, a.e);
}

// ------------------------------------------------------------------------------------------------
// wgrad : per tap, dw[tap][ci][co] += scale[co] * sum_m x[pix(m,tap)][ci] * dy[m][co]
// grid = (ci tiles * co tiles, 9 taps, row splits); reduction rows are the output pixels.
// ------------------------------------------------------------------------------------------------
struct ConvWgradArgs {
    int N, Hi, Wi, Ci, Ho, Wo, Co, stride, pad;
    const float *x;
    const float *dy;
    float *dw;
    int M, rows_per_split;
    long long part_stride;   // 9*Ci*Co when the splits write partial slabs, 0 for the atomic fallback
    int tiles_m, tiles_n;
    EpiArgs e;
    int s16;                 // x and dy are bf16 in memory (bf16 activation storage; bf16 compute only)
    int slab_ts;             // partial slabs in tile order (gemm_core.h: store_slab_ts; tile = (tap * tiles_m + tm) * tiles_n + tn)
};

}  // namespace detr
#include "conv_x3.h"
namespace detr {

template <int BM>
struct LoaderWgradA {
    static constexpr int VPR = BM / 4;
    static constexpr int TOTAL = GEMM_BK * VPR;
    static constexpr int NV = (TOTAL >= GEMM_THREADS) ? TOTAL / GEMM_THREADS : 1;
    BufSrc src;
    int n_[NV], h_[NV], w_[NV];  // output-pixel coordinates of this entry's current reduction row
    int m_[NV];
    int tid, ci0, kh, kw;

    __device__ __forceinline__ void init(const ConvWgradArgs &a, int ci0_, int tap, int m_begin, int tid_) {
        src.init(a.x, (long long)a.N * a.Hi * a.Wi * a.Ci);
        tid = tid_;
        ci0 = ci0_;
        kh = tap / 3;
        kw = tap - kh * 3;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + GEMM_THREADS * i;
            const int kr = idx / VPR;
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
            const int idx = tid + GEMM_THREADS * i;
            const int c4 = (idx % VPR) * 4;
            const int hs = h_[i] * a.stride - a.pad + kh;
            const int ws = w_[i] * a.stride - a.pad + kw;
            const bool v = (idx < TOTAL) && (m_[i] < m_end) && hs >= 0 && ws >= 0 && hs < a.Hi && ws < a.Wi &&
                           (ci0 + c4 < a.Ci);
            const unsigned off = ((unsigned)((n_[i] * a.Hi + hs) * a.Wi + ws) * (unsigned)a.Ci + (unsigned)(ci0 + c4)) * 4u;
            r[i] = src.ld4(v ? off : BUF_OOB);
        }
    }
    __device__ __forceinline__ void advance(const ConvWgradArgs &a) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            m_[i] += GEMM_BK;
            w_[i] += GEMM_BK;
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
    template <int LD>
    __device__ __forceinline__ void store(float (*S)[LD], const float4 (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + GEMM_THREADS * i;
            if (idx < TOTAL) {
                const int kr = idx / VPR;
                const int c4 = (idx % VPR) * 4;
                *reinterpret_cast<float4 *>(&S[kr][c4]) = r[i];
            }
        }
    }
};

template <int BM, int BN, int WGM, int WGN, bool SPLIT3 = false>
__global__ __launch_bounds__(GEMM_THREADS) void conv3x3_wgrad_kernel(ConvWgradArgs a) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    __shared__ __attribute__((aligned(16))) char smem_raw[SmemBytes<BM, BN, WGN>::VALUE];
    GemmSmem<BM, BN> &sm = *reinterpret_cast<GemmSmem<BM, BN> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tn = blockIdx.x % a.tiles_n, tm = blockIdx.x / a.tiles_n;
    const int ci0 = tm * BM, co0 = tn * BN;
    const int tap = blockIdx.y;
    const int m_begin = blockIdx.z * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    if (m_begin >= m_end) return;
    const int nkt = (m_end - m_begin + GEMM_BK - 1) / GEMM_BK;

    LoaderWgradA<BM> la;
    la.init(a, ci0, tap, m_begin, tid);
    LoaderMN<BN> lb;
    lb.init(a.dy, a.Co, co0, a.Co, a.M, true, tid);

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[LoaderWgradA<BM>::NV], rb[LoaderMN<BN>::NV];
    la.load(a, m_end, ra);
    lb.load(m_begin, m_end, rb);
    la.store(sm.A[0], ra);
    lb.store(sm.B[0], rb);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (more) {
            la.advance(a);
            la.load(a, m_end, ra);
            lb.load(m_begin + (kt + 1) * GEMM_BK, m_end, rb);
        }
        mma_ktile_sel<BM, BN, WGM, WGN, SPLIT3>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
        if (more) {
            la.store(sm.A[cur ^ 1], ra);
            lb.store(sm.B[cur ^ 1], rb);
        }
        __syncthreads();
        cur ^= 1;
    }
    if constexpr (WGM == 2 && WGN == 2) {
        if (a.slab_ts) {
            float *slab = a.dw + (long long)blockIdx.z * a.part_stride + ((long long)(tap * a.tiles_m + tm) * a.tiles_n + tn) * (BM * BN);
            store_slab_ts<BM, BN, WGM, WGN>(acc, slab, wave, lane);
            return;
        }
    }
    float *dw = a.dw + (long long)tap * a.Ci * a.Co + (long long)blockIdx.z * a.part_stride;
    epilogue<BM, BN, WGM, WGN, false>(acc, reinterpret_cast<float *>(smem_raw), dw, a.Co, a.Ci, a.Co, ci0, co0, wm, wn, lane, wave, a.e);
}

// ------------------------------------------------------------------------------------------------
// bf16-compute variants (fp32 storage, bf16 MFMA; see gemm_bf16_core.h).  Channel counts must be % 32.
// ------------------------------------------------------------------------------------------------
// X16: the source tensor is bf16 in memory (bf16 activation storage): 16-byte chunks of 8 channels, no conversion
template <int BM, bool DGRAD, bool X16>
struct LoaderConvAb {
    static constexpr int NV = X16 ? BM / 64 : BM / 32;
    typedef typename std::conditional<X16, uint4, float4>::type Reg;
    BufSrc src;
    int n_[NV], h_[NV], w_[NV];
    bool ok[NV];
    int kq, tid;

    __device__ __forceinline__ void init(const ConvArgs &a, int m0, int tid_) {
        src.init_bytes(a.src, (long long)a.N * a.Hs * a.Ws * a.Cs * (X16 ? 2 : 4));
        tid = tid_;
        kq = X16 ? (tid & 3) * 8 : (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int m = m0 + (X16 ? (tid >> 2) + 64 * i : (tid >> 3) + 32 * i);
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
    // live = false: a request past the last K tile (operand pipeline tail) -- every lane takes the out-of-range offset
    __device__ __forceinline__ void load(const ConvArgs &a, int kt, int cpt, Reg (&r)[NV], bool live = true) const {
        const int tap = kt / cpt;
        int kh, kw;
        conv_tap(a, tap, kh, kw);
        load_tap(a, kh, kw, kt - tap * cpt, r, live);
    }
    // the same with the tap (kh, kw) and the channel tile kc inside the tap given by the caller (the main loop keeps them
    // as running counters: the divisions of load() are ~40 scalar instructions per K tile, more than the tile's MFMAs)
    __device__ __forceinline__ void load_tap(const ConvArgs &a, int kh, int kw, int kc, Reg (&r)[NV], bool live = true) const {
        const int c0 = kc * BF_BK + kq;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int hs, ws;
            bool v = ok[i] && live;
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
            const unsigned off = ((unsigned)((n_[i] * a.Hs + hs) * a.Ws + ws) * (unsigned)a.Cs + (unsigned)c0) * (X16 ? 2u : 4u);
            if constexpr (X16) r[i] = src.ld16(v ? off : BUF_OOB);
            else r[i] = src.ld4(v ? off : BUF_OOB);
        }
    }
    __device__ __forceinline__ void store(unsigned short (*S)[BF_LD], const Reg (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (X16) {
                *reinterpret_cast<uint4 *>(&S[(tid >> 2) + 64 * i][kq]) = r[i];
            } else {
                const int row = (tid >> 3) + 32 * i;
                *reinterpret_cast<uint2 *>(&S[row][kq]) = make_uint2(pack_bf16(r[i].x, r[i].y), pack_bf16(r[i].z, r[i].w));
            }
        }
    }
};

template <int BM, int BN, int WGM, int WGN, bool DGRAD, bool W16, bool X16>
__global__ __launch_bounds__(GEMM_THREADS, (BM * BN >= 128 * 128) ? 3 : (BM * BN == 64 * 64 ? DETR_GEMM64_MINW : 1)) void conv3x3_bf16c_kernel(ConvArgs a) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    __shared__ __attribute__((aligned(16))) char smem_raw[BfSmemBytes<BM, BN, WGN>::VALUE];
    BfSmem<BM, BN> &sm = *reinterpret_cast<BfSmem<BM, BN> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = id % a.tiles_n, tm = id / a.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int cpt = a.Cs / BF_BK;
    const int nkt = (a.par_on ? a.nth * a.ntw : 9) * cpt;
    const long long tapstride = (long long)a.Ci * a.Co;
    LoaderConvAb<BM, DGRAD, X16> la;
    la.init(a, m0, tid);
    // fwd weights: transpose-read image; W16: the kernel is already bf16 in memory (per-step weight shadow)
    using LB = typename std::conditional<W16, typename std::conditional<DGRAD, LoaderKh<BN>, LoaderMNth<BN, true>>::type,
                                         typename std::conditional<DGRAD, LoaderKb<BN>, LoaderMNt<BN>>::type>::type;
    constexpr int NRB = LB::NREG;
    LB lb;
    lb.init(a.w, a.Co, n0, a.Cd, a.Cs, true, tid, 9 * tapstride);   // one descriptor over the 9 taps
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // Operand pipeline two K tiles deep with LDS-only barriers -- see gemm_bf16c_body (gemm_f32.hip): requests and LDS
    // stores are unconditional (past the last tile they resolve to the out-of-range offset / write a buffer nobody reads)
    // so that the compiler's vmcnt waits stay exact.
    using LAc = LoaderConvAb<BM, DGRAD, X16>;
    typename LAc::Reg ra0[LAc::NV], ra1[LAc::NV];
    typename LB::Reg rb0[NRB], rb1[NRB];
    // K tiles are requested strictly in order (0, 1, 2, then kt + 3 per iteration): the tap and the channel tile inside it
    // are running counters instead of kt / cpt and conv_tap()'s divisions
    const int tap_cols = a.par_on ? a.ntw : 3;
    int it_kc = 0, it_ti = 0, it_tj = 0;
    auto load_ab = [&](int kt, typename LAc::Reg (&ra)[LAc::NV], typename LB::Reg (&rb)[NRB]) {
        const bool live = kt < nkt;
        const int kh = a.par_on ? a.kh0 + 2 * it_ti : it_ti;
        const int kw = a.par_on ? a.kw0 + 2 * it_tj : it_tj;
        la.load_tap(a, kh, kw, it_kc, ra, live);
        lb.load(it_kc * BF_BK, live ? a.Cs : 0, rb, live ? (unsigned)((kh * 3 + kw) * tapstride * (W16 ? 2 : 4)) : 0u);
        if (++it_kc == cpt) {
            it_kc = 0;
            if (++it_tj == tap_cols) {
                it_tj = 0;
                ++it_ti;
            }
        }
    };
    load_ab(0, ra0, rb0);
    la.store(sm.A[0], ra0);
    lb.store(sm.B[0], rb0);
    load_ab(1, ra0, rb0);
    load_ab(2, ra1, rb1);
    lds_barrier();
    auto iter = [&](const int kt, const int cur, typename LAc::Reg (&rpa)[LAc::NV], typename LB::Reg (&rpb)[NRB]) {
        if constexpr ((DETR_ABLATE & 4) == 0) {
            la.store(sm.A[cur ^ 1], rpa);
            lb.store(sm.B[cur ^ 1], rpb);
        } else {
            for (int i = 0; i < LAc::NV; ++i) ablate_keep(rpa[i]);
            for (int i = 0; i < NRB; ++i) ablate_keep(rpb[i]);
        }
        if constexpr ((DETR_ABLATE & 2) == 0) load_ab(kt + 3, rpa, rpb);
        mma_ktile_bf16<BM, BN, WGM, WGN, false, !DGRAD>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
        if constexpr ((DETR_ABLATE & 8) == 0) lds_barrier();
    };
    {
        int kt = 0;
        for (; kt + 2 <= nkt; kt += 2) {
            iter(kt, 0, ra0, rb0);
            iter(kt + 1, 1, ra1, rb1);
        }
        if (kt < nkt) iter(kt, 0, ra0, rb0);
    }
    __syncthreads();
    epilogue<BM, BN, WGM, WGN>(acc, reinterpret_cast<float *>(smem_raw), a.dst, a.Cd, a.M, a.Cd, m0, n0, wm, wn, lane, wave, a.e);
}

// wgrad: A'[i = ci][k = m] gathered into a transpose-read image (gemm_bf16_core.h LoaderMNt): unit u = t + 256*i holds
// the float4 of channels ci0 + 16*ib + 4*c of reduction row (output pixel) k = 4*kb + kr of the tile.
template <int BM, bool S16>
struct LoaderWgradAt {
    static constexpr int NB = BM / 16;
    static constexpr int NU = BM / 32;
    typedef typename std::conditional<S16, uint2, float4>::type Reg;
    BufSrc src;
    int n_[NU], h_[NU], w_[NU], m_[NU];
    int tid, ci0, kh, kw;

    __device__ __forceinline__ void init(const ConvWgradArgs &a, int ci0_, int tap, int m_begin, int tid_) {
        src.init_bytes(a.x, (long long)a.N * a.Hi * a.Wi * a.Ci * (S16 ? 2 : 4));
        tid = tid_; ci0 = ci0_;
        kh = tap / 3; kw = tap - kh * 3;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + 256 * i;
            const int m = m_begin + 4 * (u / (16 * NB)) + ((u >> 2) & 3);
            m_[i] = m;
            w_[i] = m % a.Wo;
            const int t = m / a.Wo;
            h_[i] = t % a.Ho;
            n_[i] = t / a.Ho;
        }
    }
    __device__ __forceinline__ void load(const ConvWgradArgs &a, int m_end, Reg (&r)[NU]) const {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + 256 * i;
            const int c = ci0 + 16 * ((u >> 4) & (NB - 1)) + 4 * (u & 3);
            const int hs = h_[i] * a.stride - a.pad + kh, ws = w_[i] * a.stride - a.pad + kw;
            const bool v = m_[i] < m_end && hs >= 0 && ws >= 0 && hs < a.Hi && ws < a.Wi && c < a.Ci;
            const unsigned off = ((unsigned)((n_[i] * a.Hi + hs) * a.Wi + ws) * (unsigned)a.Ci + (unsigned)c) * (S16 ? 2u : 4u);
            if constexpr (S16) r[i] = src.ld8(v ? off : BUF_OOB);
            else r[i] = src.ld4(v ? off : BUF_OOB);
        }
    }
    __device__ __forceinline__ void advance(const ConvWgradArgs &a) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            m_[i] += BF_BK;
            w_[i] += BF_BK;
            while (w_[i] >= a.Wo) { w_[i] -= a.Wo; h_[i] += 1; }
            while (h_[i] >= a.Ho) { h_[i] -= a.Ho; n_[i] += 1; }
        }
    }
    __device__ __forceinline__ void store(unsigned short (*S)[BF_LD], const Reg (&r)[NU]) const {
        unsigned short *flat = &S[0][0];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            if constexpr (S16) *reinterpret_cast<uint2 *>(flat + (tid + 256 * i) * 4) = r[i];
            else *reinterpret_cast<uint2 *>(flat + (tid + 256 * i) * 4) = make_uint2(pack_bf16(r[i].x, r[i].y), pack_bf16(r[i].z, r[i].w));
        }
    }
};

template <int BM, int BN, int WGM, int WGN, bool S16>
__global__ __launch_bounds__(GEMM_THREADS, (BM * BN >= 128 * 128) ? 3 : 1) void conv3x3_wgrad_bf16c_kernel(ConvWgradArgs a) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    __shared__ __attribute__((aligned(16))) char smem_raw[BfSmemBytes<BM, BN, WGN>::VALUE];
    BfSmem<BM, BN> &sm = *reinterpret_cast<BfSmem<BM, BN> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tn = blockIdx.x % a.tiles_n, tm = blockIdx.x / a.tiles_n;
    const int ci0 = tm * BM, co0 = tn * BN;
    const int tap = blockIdx.y;
    const int m_begin = blockIdx.z * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    if (m_begin >= m_end) return;
    const int nkt = (m_end - m_begin + BF_BK - 1) / BF_BK;
    LoaderWgradAt<BM, S16> la;
    la.init(a, ci0, tap, m_begin, tid);
    typename std::conditional<S16, LoaderMNth<BN>, LoaderMNt<BN>>::type lb;      // dy: same storage type as x
    lb.init(a.dy, a.Co, co0, a.Co, a.M, true, tid);
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    typename LoaderWgradAt<BM, S16>::Reg ra[LoaderWgradAt<BM, S16>::NU];
    typename std::conditional<S16, uint2, float4>::type rb[LoaderMNt<BN>::NU];
    la.load(a, m_end, ra);
    lb.load(m_begin, m_end, rb);
    la.store(sm.A[0], ra);
    lb.store(sm.B[0], rb);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (more) {
            la.advance(a);
            la.load(a, m_end, ra);
            lb.load(m_begin + (kt + 1) * BF_BK, m_end, rb);
        }
        mma_ktile_bf16<BM, BN, WGM, WGN, true, true>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
        if (more) {
            la.store(sm.A[cur ^ 1], ra);
            lb.store(sm.B[cur ^ 1], rb);
        }
        __syncthreads();
        cur ^= 1;
    }
    if constexpr (WGM == 2 && WGN == 2) {
        if (a.slab_ts) {
            float *slab = a.dw + (long long)blockIdx.z * a.part_stride + ((long long)(tap * a.tiles_m + tm) * a.tiles_n + tn) * (BM * BN);
            store_slab_ts<BM, BN, WGM, WGN>(acc, slab, wave, lane);
            return;
        }
    }
    float *dw = a.dw + (long long)tap * a.Ci * a.Co + (long long)blockIdx.z * a.part_stride;
    epilogue<BM, BN, WGM, WGN, false>(acc, reinterpret_cast<float *>(smem_raw), dw, a.Co, a.Ci, a.Co, ci0, co0, wm, wn, lane, wave, a.e);
}

// ------------------------------------------------------------------------------------------------
// bf16 weight gradient with the NINE TAPS FUSED (stride 1, pad 1, Ci % 64 == 0, Co % 64 == 0):
//   dw[tap][ci][co] += sum over output pixels of x[pix + tap][ci] * dy[pix][co]
// One workgroup owns a 64(ci) x 64(co) tile of all nine taps (9 accumulators of 16 registers per wave).  The reduction
// runs over UNITS of 32 consecutive output pixels of one output row: per unit the workgroup stages the dy tile
// [32 px][64 co] and the haloed input patch [3 rows][34 px][64 ci] ONCE (bf16, natural [pixel][channel] orientation
// cut into [4 px][16 ch] sub-blocks of 128 B) and every tap reads its MFMA fragments with ds_read_b64_tr_b16 at a
// shifted pixel row -- the per-tap kernel above re-loads x and dy for every tap (4.2x the LDS fill per MFMA).
// ------------------------------------------------------------------------------------------------
// STR = 2 (round 5; bf16 storage only): the stride-2 convolutions of layer2-4 (resnet_backbone.py:123-126 with strides 2) took the per-tap kernel
// until now (100 us per launch against 63 for the same FLOPs at stride 1).  A unit of 32 output pixels reads input columns 2 j + kw - 1: the
// patch is 3 rows x 65 pixels, staged DE-INTERLEAVED -- even patch columns in slots 0..32, odd ones in slots 33..65 -- so that the tap kw of
// output pixel j sits at slot (kw & 1) * 33 + j + (kw >> 1): consecutive slots for consecutive j, the same transpose-read fragments as at stride 1.
template <int STR>
struct WgradFusedGeom {
    static constexpr int PC = STR == 1 ? 34 : 66;                 // pixel slots of a patch row (stride 2: 33 even + 33 odd columns, the last odd one unused)
    static constexpr int NG = (PC + 3) / 4;                       // pixel groups of 4
    static constexpr int ROW = NG * 4 * 64;                       // shorts per patch row: pixel groups x 4 ci blocks x [4][16]
    __device__ static constexpr int slot(int c) { return STR == 1 ? c : (c & 1) * 33 + (c >> 1); }
};
constexpr int WF_ROW = WgradFusedGeom<1>::ROW;

template <int STR>
struct WgradFusedSmem {
    unsigned short X[2][3 * WgradFusedGeom<STR>::ROW];      // haloed input patch
    unsigned short D[2][32 * 64];         // dy tile (transpose-read image, BMN = 64)
};

template <bool S16, int STR = 1>
__global__ __launch_bounds__(GEMM_THREADS, 2) void conv3x3_wgrad_fused_bf16_kernel(ConvWgradArgs a, int units_per_split,
                                                                                 int chunks) {
    static_assert(STR == 1 || S16, "the stride-2 form exists for bf16-stored tensors");
    using WG = WgradFusedGeom<STR>;
    constexpr int WF_ROW = WG::ROW, PC = WG::PC, NPP = 3 * PC;
    constexpr int SMEM = (int)sizeof(WgradFusedSmem<STR>) > SmemBytes<64, 64, 2>::VALUE ? (int)sizeof(WgradFusedSmem<STR>)
                                                                                         : SmemBytes<64, 64, 2>::VALUE;
    __shared__ __attribute__((aligned(16))) char smem_raw[SMEM];
    WgradFusedSmem<STR> &sm = *reinterpret_cast<WgradFusedSmem<STR> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tn = blockIdx.x % a.tiles_n, tm = blockIdx.x / a.tiles_n;
    const int ci0 = tm * 64, co0 = tn * 64;
    const int total_units = a.N * a.Ho * chunks;
    const int u_begin = blockIdx.z * units_per_split;
    const int u_end = min(total_units, u_begin + units_per_split);
    if (u_begin >= u_end) return;
    BufSrc xs, ds;
    xs.init_bytes(a.x, (long long)a.N * a.Hi * a.Wi * a.Ci * (S16 ? 2 : 4));
    ds.init_bytes(a.dy, (long long)a.N * a.Ho * a.Wo * a.Co * (S16 ? 2 : 4));

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // S16: 16-byte granules (8 channels of one pixel; two adjacent 8-byte units of the transpose-read images) -- 5 requests
    // and 5 ds_write_b128 per thread and unit instead of 9 + 9 eight-byte ones; fp32 storage: float4 = 4 channels
    typedef typename std::conditional<S16, uint4, float4>::type Reg;
    constexpr int NRX = (NPP * (S16 ? 8 : 16) + 255) / 256, NRD = S16 ? 1 : 2;      // stride 1: 4 (bf16) / 7 (fp32); stride 2: 7
    // two register sets: the operand pipeline is two units deep (see gemm_bf16c_body); requests past u_end take the
    // out-of-range offset and every unit is stored, so that the vmcnt waits in front of the LDS stores stay exact
    Reg rx0[NRX], rd0[NRD], rx1[NRX], rd1[NRD];
    // units are requested strictly in order (u_begin, +1, +2, ...): (chunk, output row, image) are running counters
    // instead of three divisions per unit
    int it_chunk = u_begin % chunks, it_ho = (u_begin / chunks) % a.Ho, it_n = (u_begin / chunks) / a.Ho;
    auto load_unit = [&](int u, Reg (&rx)[NRX], Reg (&rd)[NRD]) {
        const bool live = u < u_end;
        const int ho = it_ho, n = it_n;
        const int wo0 = it_chunk * 32;
        if (++it_chunk == chunks) {
            it_chunk = 0;
            if (++it_ho == a.Ho) {
                it_ho = 0;
                ++it_n;
            }
        }
        // dy tile: LoaderMNt<64> map (row = pixel j, col = co); S16: unit pairs (2 v, 2 v + 1) = 8 channels
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
            const int v = S16 ? 2 * tid : tid + 256 * i;
            const int j = 4 * (v >> 6) + ((v >> 2) & 3);
            const int col = 16 * ((v >> 4) & 3) + 4 * (v & 3);
            const bool ok = live && wo0 + j < a.Wo;
            const unsigned off = ((unsigned)((n * a.Ho + ho) * a.Wo + wo0 + j) * (unsigned)a.Co + (unsigned)(co0 + col)) * (S16 ? 2u : 4u);
            if constexpr (S16) rd[i] = ds.ld16(ok ? off : BUF_OOB);
            else rd[i] = ds.ld4(ok ? off : BUF_OOB);
        }
        // input patch: slot v -> (patch pixel pp in [0, 3 PC), channel granule): S16 8 granules of 8 channels, fp32 16 of 4
#pragma unroll
        for (int i = 0; i < NRX; ++i) {
            const int v = tid + 256 * i;
            const int pp = S16 ? v >> 3 : v >> 4, c4 = S16 ? 2 * (v & 7) : v & 15;
            const int kh = pp / PC, c = pp - kh * PC;
            const int hi = STR * ho - 1 + kh, wi = STR * wo0 - 1 + c;
            const bool ok = live && pp < NPP && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi;
            const unsigned off = ((unsigned)((n * a.Hi + hi) * a.Wi + wi) * (unsigned)a.Ci + (unsigned)(ci0 + 4 * c4)) * (S16 ? 2u : 4u);
            if constexpr (S16) rx[i] = xs.ld16(ok ? off : BUF_OOB);
            else rx[i] = xs.ld4(ok ? off : BUF_OOB);
        }
    };
    auto store_unit = [&](int buf, const Reg (&rx)[NRX], const Reg (&rd)[NRD]) {
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
            if constexpr (S16) *reinterpret_cast<uint4 *>(&sm.D[buf][tid * 8]) = rd[i];
            else *reinterpret_cast<uint2 *>(&sm.D[buf][(tid + 256 * i) * 4]) = make_uint2(pack_bf16(rd[i].x, rd[i].y), pack_bf16(rd[i].z, rd[i].w));
        }
#pragma unroll
        for (int i = 0; i < NRX; ++i) {
            const int v = tid + 256 * i;
            const int pp = S16 ? v >> 3 : v >> 4, c4 = S16 ? 2 * (v & 7) : v & 15;
            if (pp < NPP) {
                const int kh = pp / PC, c = WG::slot(pp - kh * PC);
                // (round 6) pixel (c & 3) of ci block cb sits in row ((c & 3) + cb) & 3 of its [4 px][16 ch] sub-block: the four blocks of one pixel -- the
                // 8 (bf16) / 16 (fp32) lanes of one store group -- used to land on the same 8 banks of 32, a 4-way conflict on every patch store
                // (SQ_LDS_BANK_CONFLICT 0.16 of the CU's cycles, rounds 3-5); the transpose read only cares which LANE hands in which chunk
                const int o = kh * WF_ROW + ((c >> 2) * 4 + (c4 >> 2)) * 64 + (((c & 3) + (c4 >> 2)) & 3) * 16 + (c4 & 3) * 4;
                if constexpr (S16) *reinterpret_cast<uint4 *>(&sm.X[buf][o]) = rx[i];
                else *reinterpret_cast<uint2 *>(&sm.X[buf][o]) = make_uint2(pack_bf16(rx[i].x, rx[i].y), pack_bf16(rx[i].z, rx[i].w));
            }
        }
    };
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const int g = lane >> 4, t16 = lane & 15;
    // fragment offsets (shorts) inside one buffer: they do not depend on the unit, the tap row kh is an immediate offset
    int xo[2][3][2], dofs[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int cib = 2 * wm + (g & 1);                        // 16-channel block of this lane group
        const int jbase = 16 * s2 + 8 * (g >> 1) + (t16 >> 2);   // output pixel handed in by this lane (read 1)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = STR == 1 ? jbase + kw + 4 * h : (kw & 1) * 33 + jbase + (kw >> 1) + 4 * h;
                xo[s2][kw][h] = ((c >> 2) * 4 + cib) * 64 + (((c & 3) + cib) & 3) * 16 + (t16 & 3) * 4;      // (rotated rows: see store_unit)
            }
        dofs[s2] = ((4 * s2 + 2 * (g >> 1)) * 4 + 2 * wn + (g & 1)) * 64 + t16 * 4;
    }
    load_unit(u_begin, rx0, rd0);
    store_unit(0, rx0, rd0);
    load_unit(u_begin + 1, rx0, rd0);
    load_unit(u_begin + 2, rx1, rd1);
    lds_barrier();
    // one unit: the set `rp` holds unit u+1 (stored now, then refilled with unit u+3); LDS buffer cur holds unit u
    auto iter = [&](const int u, const int cur, Reg (&rpx)[NRX], Reg (&rpd)[NRD]) {
        if constexpr ((DETR_ABLATE & 4) == 0) store_unit(cur ^ 1, rpx, rpd);
        else { for (int i = 0; i < NRX; ++i) ablate_keep(rpx[i]); for (int i = 0; i < NRD; ++i) ablate_keep(rpd[i]); }
        if constexpr ((DETR_ABLATE & 2) == 0) load_unit(u + 3, rpx, rpd);
        const unsigned short *X = sm.X[cur];
        const unsigned short *D = sm.D[cur];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            // B fragment: dy[pixels 16 s2 + 8 hi ..][co = wn*32 + (lane & 31)]  (image of LoaderMNt<64>)
            bf16x8 bfrag;
            if constexpr ((DETR_ABLATE & 16) != 0) bfrag = __builtin_bit_cast(bf16x8, make_uint4(lane, s2, 1, 0x3f803f80u));
            else {
                const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(D + dofs[s2]));
                const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(D + dofs[s2] + 4 * 64));
                bfrag = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    bf16x8 afrag;
                    if constexpr ((DETR_ABLATE & 16) != 0) afrag = __builtin_bit_cast(bf16x8, make_uint4(lane, kh, kw, 0x3f803f80u));
                    else {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(X + xo[s2][kw][0] + kh * WF_ROW));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(X + xo[s2][kw][1] + kh * WF_ROW));
                        afrag = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                    if constexpr ((DETR_ABLATE & 1) != 0) { ablate_keep(afrag); ablate_keep(bfrag); }
                    else acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag, bfrag, acc[kh * 3 + kw], 0, 0, 0);
                }
        }
        if constexpr ((DETR_ABLATE & 8) == 0) lds_barrier();
    };
    {
        int u = u_begin;
        for (; u + 2 <= u_end; u += 2) {
            iter(u, 0, rx0, rd0);
            iter(u + 1, 1, rx1, rd1);
        }
        if (u < u_end) iter(u, 0, rx0, rd0);
    }
    if (a.slab_ts) {       // nine 64 x 64 tiles in accumulator-register order: 36 lane-linear 16-byte stores per lane, no LDS
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f32x16 one[1][1];
            one[0][0] = acc[t];
            float *slab = a.dw + (long long)blockIdx.z * a.part_stride + ((long long)(t * a.tiles_m + tm) * a.tiles_n + tn) * (64 * 64);
            store_slab_ts<64, 64, 2, 2>(one, slab, wave, lane);
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        f32x16 one[1][1];
        one[0][0] = acc[t];
        float *dw = a.dw + (long long)t * a.Ci * a.Co + (long long)blockIdx.z * a.part_stride;
        epilogue<64, 64, 2, 2, false>(one, reinterpret_cast<float *>(smem_raw), dw, a.Co, a.Ci, a.Co, ci0, co0, wm, wn, lane, wave, a.e, t == 0);
    }
}

// ------------------------------------------------------------------------------------------------
// stem / pooling / subsample elementwise kernels
// ------------------------------------------------------------------------------------------------
__global__ void maxpool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, uint8_t *__restrict__ amax,
                                   int N, int H, int W, int C, int Ho, int Wo, long long total) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        long long t = idx / C;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
                float v = 0.0f;  // the explicit ZeroPadding2D(1) takes part in the max
                if (hi >= 0 && wi >= 0 && hi < H && wi < W) v = x[(((long long)n * H + hi) * W + wi) * C + c];
                if (v > best) {
                    best = v;
                    bi = kh * 3 + kw;
                }
            }
        y[idx] = best;
        amax[idx] = (uint8_t)bi;
    }
}

// dx[n,h,w,c] = (x > 0) * sum over the (<= 4) pooling windows that contain (h,w) and selected it.
// Four channels per thread: float4 x / dy / dx accesses, one 32-bit load of the four argmax bytes.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ amax,
                                                          const float *__restrict__ x, float *__restrict__ dx, int N, int H,
                                                          int W, int C, int Ho, int Wo, long long total4) {
    const int C4 = C >> 2;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total4;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const unsigned pix = (unsigned)(idx / C4);
        const int w = (int)(pix % (unsigned)W);
        const unsigned t = pix / (unsigned)W;
        const int h = (int)(t % (unsigned)H);
        const int n = (int)(t / (unsigned)H);
        const float4 xv = reinterpret_cast<const float4 *>(x)[idx];
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int th = h + 1 - kh;
            if (th < 0 || (th & 1)) continue;
            const int ho = th >> 1;
            if (ho >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tw = w + 1 - kw;
                if (tw < 0 || (tw & 1)) continue;
                const int wo = tw >> 1;
                if (wo >= Wo) continue;
                const long long o4 = ((long long)(n * Ho + ho) * Wo + wo) * C4 + c4;
                const uint32_t am = reinterpret_cast<const uint32_t *>(amax)[o4];
                const float4 d = reinterpret_cast<const float4 *>(dy)[o4];
                const uint32_t tap = (uint32_t)(kh * 3 + kw);
                if ((am & 0xFFu) == tap) g.x += d.x;
                if (((am >> 8) & 0xFFu) == tap) g.y += d.y;
                if (((am >> 16) & 0xFFu) == tap) g.z += d.z;
                if ((am >> 24) == tap) g.w += d.w;
            }
        }
        g.x = xv.x > 0.0f ? g.x : 0.0f;
        g.y = xv.y > 0.0f ? g.y : 0.0f;
        g.z = xv.z > 0.0f ? g.z : 0.0f;
        g.w = xv.w > 0.0f ? g.w : 0.0f;
        reinterpret_cast<float4 *>(dx)[idx] = g;
    }
}

// bf16-storage twins of the two pooling kernels (4 channels = 8 bytes per thread)
__device__ __forceinline__ float4 bf4_to_f4(uint2 v) {
    return make_float4(bf16_bits_to_f32(v.x & 0xFFFFu), bf16_bits_to_f32(v.x >> 16), bf16_bits_to_f32(v.y & 0xFFFFu), bf16_bits_to_f32(v.y >> 16));
}
__global__ __launch_bounds__(256) void maxpool_fwd_bf16_kernel(const uint2 *__restrict__ x, uint2 *__restrict__ y,
                                                               uint32_t *__restrict__ amax, int N, int H, int W, int C4,
                                                               int Ho, int Wo, long long total4) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total4;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const unsigned pix = (unsigned)(idx / C4);
        const int wo = (int)(pix % (unsigned)Wo);
        const unsigned t = pix / (unsigned)Wo;
        const int ho = (int)(t % (unsigned)Ho);
        const int n = (int)(t / (unsigned)Ho);
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        uint32_t bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);   // the explicit ZeroPadding2D(1) takes part in the max
                if (hi >= 0 && wi >= 0 && hi < H && wi < W) v = bf4_to_f4(x[((long long)(n * H + hi) * W + wi) * C4 + c4]);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (vv[e] > best[e]) { best[e] = vv[e]; bi[e] = (uint32_t)(kh * 3 + kw); }
            }
        y[idx] = make_uint2(f32_to_bf16_pair(best[0], best[1]), f32_to_bf16_pair(best[2], best[3]));   // exact: inputs are bf16
        amax[idx] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_bf16_kernel(const uint2 *__restrict__ dy, const uint32_t *__restrict__ amax,
                                                               const uint2 *__restrict__ x, uint2 *__restrict__ dx, int N, int H,
                                                               int W, int C4, int Ho, int Wo, long long total4) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total4;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const unsigned pix = (unsigned)(idx / C4);
        const int w = (int)(pix % (unsigned)W);
        const unsigned t = pix / (unsigned)W;
        const int h = (int)(t % (unsigned)H);
        const int n = (int)(t / (unsigned)H);
        const float4 xv = bf4_to_f4(x[idx]);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int th = h + 1 - kh;
            if (th < 0 || (th & 1)) continue;
            const int ho = th >> 1;
            if (ho >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tw = w + 1 - kw;
                if (tw < 0 || (tw & 1)) continue;
                const int wo = tw >> 1;
                if (wo >= Wo) continue;
                const long long o4 = ((long long)(n * Ho + ho) * Wo + wo) * C4 + c4;
                const uint32_t am = amax[o4];
                const float4 d = bf4_to_f4(dy[o4]);
                const uint32_t tap = (uint32_t)(kh * 3 + kw);
                if ((am & 0xFFu) == tap) g.x += d.x;
                if (((am >> 8) & 0xFFu) == tap) g.y += d.y;
                if (((am >> 16) & 0xFFu) == tap) g.z += d.z;
                if ((am >> 24) == tap) g.w += d.w;
            }
        }
        g.x = xv.x > 0.0f ? g.x : 0.0f;
        g.y = xv.y > 0.0f ? g.y : 0.0f;
        g.z = xv.z > 0.0f ? g.z : 0.0f;
        g.w = xv.w > 0.0f ? g.w : 0.0f;
        dx[idx] = make_uint2(f32_to_bf16_pair(g.x, g.y), f32_to_bf16_pair(g.z, g.w));
    }
}
// 8-channel (16-byte) forms of the two bf16 pooling kernels: half the memory instructions and half the index arithmetic per
// byte of the 4-channel forms above (which stay for C % 8 != 0).  rocprofv3, stem map 8 x 400 x 667 x 64: backward 266 us
// at 2.3 TB/s with 8-byte accesses.
__device__ __forceinline__ void bf8_to_f8(uint4 v, float (&o)[8]) {
    o[0] = __builtin_bit_cast(float, v.x << 16); o[1] = __builtin_bit_cast(float, v.x & 0xFFFF0000u);
    o[2] = __builtin_bit_cast(float, v.y << 16); o[3] = __builtin_bit_cast(float, v.y & 0xFFFF0000u);
    o[4] = __builtin_bit_cast(float, v.z << 16); o[5] = __builtin_bit_cast(float, v.z & 0xFFFF0000u);
    o[6] = __builtin_bit_cast(float, v.w << 16); o[7] = __builtin_bit_cast(float, v.w & 0xFFFF0000u);
}

__global__ __launch_bounds__(256) void maxpool_fwd_bf16x8_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y,
                                                                 uint2 *__restrict__ amax, int N, int H, int W, int C8,
                                                                 int Ho, int Wo, long long total8) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total8;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % C8);
        const unsigned pix = (unsigned)(idx / C8);
        const int wo = (int)(pix % (unsigned)Wo);
        const unsigned t = pix / (unsigned)Wo;
        const int ho = (int)(t % (unsigned)Ho);
        const int n = (int)(t / (unsigned)Ho);
        // bf16 values compare like their bit patterns moved to the high half of a float: keep the raw winners, no re-rounding
        float best[8];
        uint32_t bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
                uint4 raw = make_uint4(0u, 0u, 0u, 0u);       // the explicit ZeroPadding2D(1) takes part in the max
                if (hi >= 0 && wi >= 0 && hi < H && wi < W) raw = x[((long long)(n * H + hi) * W + wi) * C8 + c8];
                float vv[8];
                bf8_to_f8(raw, vv);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (vv[e] > best[e]) { best[e] = vv[e]; bi[e] = (uint32_t)(kh * 3 + kw); }
            }
        y[idx] = make_uint4(f32_to_bf16_pair(best[0], best[1]), f32_to_bf16_pair(best[2], best[3]),
                            f32_to_bf16_pair(best[4], best[5]), f32_to_bf16_pair(best[6], best[7]));      // exact: inputs are bf16
        amax[idx] = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24));
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_bf16x8_kernel(const uint4 *__restrict__ dy, const uint2 *__restrict__ amax,
                                                                 const uint4 *__restrict__ x, uint4 *__restrict__ dx, int N, int H,
                                                                 int W, int C8, int Ho, int Wo, long long total8) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total8;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % C8);
        const unsigned pix = (unsigned)(idx / C8);
        const int w = (int)(pix % (unsigned)W);
        const unsigned t = pix / (unsigned)W;
        const int h = (int)(t % (unsigned)H);
        const int n = (int)(t / (unsigned)H);
        const uint4 xraw = x[idx];
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.0f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int th = h + 1 - kh;
            if (th < 0 || (th & 1)) continue;
            const int ho = th >> 1;
            if (ho >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tw = w + 1 - kw;
                if (tw < 0 || (tw & 1)) continue;
                const int wo = tw >> 1;
                if (wo >= Wo) continue;
                const long long o8 = ((long long)(n * Ho + ho) * Wo + wo) * C8 + c8;
                const uint2 am = amax[o8];
                float d[8];
                bf8_to_f8(dy[o8], d);
                const uint32_t tap = (uint32_t)(kh * 3 + kw);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (((am.x >> (8 * e)) & 0xFFu) == tap) g[e] += d[e];
                    if (((am.y >> (8 * e)) & 0xFFu) == tap) g[4 + e] += d[4 + e];
                }
            }
        }
        float xv[8];
        bf8_to_f8(xraw, xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = xv[e] > 0.0f ? g[e] : 0.0f;
        dx[idx] = make_uint4(f32_to_bf16_pair(g[0], g[1]), f32_to_bf16_pair(g[2], g[3]), f32_to_bf16_pair(g[4], g[5]), f32_to_bf16_pair(g[6], g[7]));
    }
}



__global__ void subsample2_fwd_kernel(const float4 *__restrict__ x, float4 *__restrict__ y, int H, int W, int C4,
                                      int Ho, int Wo, long long total) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long long t = idx / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long long n = t / Ho;
        y[idx] = x[((n * H + 2 * ho) * W + 2 * wo) * C4 + c];
    }
}

__global__ void subsample2_bwd_kernel(const float4 *__restrict__ dy, float4 *__restrict__ dx, int H, int W, int C4,
                                      int Ho, int Wo, long long total) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        long long t = idx / C4;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const long long n = t / H;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(h & 1) && !(w & 1) && (h >> 1) < Ho && (w >> 1) < Wo)
            v = dy[((n * Ho + (h >> 1)) * Wo + (w >> 1)) * C4 + c];
        dx[idx] = v;
    }
}

static inline int ew_grid(long long total, int block) {
    long long g = (total + block - 1) / block;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

template <int BM, int BN, int WGM, int WGN>
static void launch_conv_bf16(const ConvArgs &a0, bool dgrad, hipStream_t s) {
    ConvArgs a = a0;
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.Cd, BN);
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(GEMM_THREADS);
#define DETR_CONV_LAUNCH(W16_, X16_)                                                                                        \
    do {                                                                                                                        \
        if (dgrad) hipLaunchKernelGGL((conv3x3_bf16c_kernel<BM, BN, WGM, WGN, true, W16_, X16_>), grid, block, 0, s, a);          \
        else hipLaunchKernelGGL((conv3x3_bf16c_kernel<BM, BN, WGM, WGN, false, W16_, X16_>), grid, block, 0, s, a);                \
    } while (0)
    if (a.w16 && a.x16) DETR_CONV_LAUNCH(true, true);
    else if (a.w16) DETR_CONV_LAUNCH(true, false);
    else if (a.x16) DETR_CONV_LAUNCH(false, true);
    else DETR_CONV_LAUNCH(false, false);
#undef DETR_CONV_LAUNCH
}

template <int BM, int BN, int WGM, int WGN>
static void launch_conv(const ConvArgs &a0, bool dgrad, hipStream_t s, bool split3 = false) {
    ConvArgs a = a0;
    a.tiles_m = cdiv(a.M, BM);
    a.tiles_n = cdiv(a.Cd, BN);
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(GEMM_THREADS);
    if constexpr (BM == BN) {
        // split once into LDS (conv_x3.h): 1.0-1.25x the per-wave split below on the stride-1 / forward shapes, 0.92x on the parity classes of the
        // stride-2 input gradient (1-4 taps: short K loops), which keep the first form (profiles/r06_micro_split3.txt).  DETR_HIP_X3_CONV=2: first
        // form everywhere, =1: second form everywhere
        if (split3 && a.Cs % 32 == 0 && tune(T_X3_CONV) != 2 && (!a.par_on || tune(T_X3_CONV) == 1)) {
            if (dgrad) hipLaunchKernelGGL((conv3x3_x3_kernel<BM, BN, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv3x3_x3_kernel<BM, BN, false>), grid, block, 0, s, a);
            return;
        }
        if (split3) {
            if (dgrad) hipLaunchKernelGGL((conv3x3_kernel<BM, BN, WGM, WGN, true, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((conv3x3_kernel<BM, BN, WGM, WGN, false, true>), grid, block, 0, s, a);
            return;
        }
    }
    if (dgrad) hipLaunchKernelGGL((conv3x3_kernel<BM, BN, WGM, WGN, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_kernel<BM, BN, WGM, WGN, false>), grid, block, 0, s, a);
}

// row splits of the per-tap weight gradient (tiles = (Ci / BM) * (Co / BN) output tiles per tap) and of the fused nine-tap one
static int wgrad_split_plan(int M, int tiles, int split, int &rps) {
    if (split <= 0) {
        split = cdiv(1536, tiles * 9);
        const int max_split = cdiv(M, 256);
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    rps = cdiv(M, split);
    rps = ((rps + BF_BK - 1) / BF_BK) * BF_BK;       // multiple of both K tiles (16 and 32)
    return cdiv(M, rps);
}
// f32x3 nine-tap-per-launch weight gradient (conv3x3_wgrad_x3_kernel): 512 (128 x 128 tiles, two workgroups per CU) or 1024 (64 x 64 tiles, four) slots.
// The exact kernel's plan above (1536 workgroups, rounded UP) lands just over three rounds of 512 -- 1548 / 1539 / 1584 workgroups for 256 / 128 / 512
// channels.  Whole rounds, rounded DOWN; two rounds measured best over the five 3x3 shapes (1 / 2 / 3: within 5 % of each other; DETR_HIP_X3_WG_ROUNDS)
static int wgrad_x3_split(int M, int tiles, int bm) {
    const int rounds = tune(T_X3_WG_ROUNDS) > 0 ? tune(T_X3_WG_ROUNDS) : 2;
    const int slots = bm >= 128 ? 512 : 1024;
    int split = (rounds * slots) / (tiles * 9);
    const int max_split = cdiv(M, 256);
    if (split > max_split) split = max_split;
    return split < 1 ? 1 : split;
}

static int wgrad_fused_split_plan(int units, int tiles, int split, int &ups) {
    if (split <= 0) {
        int wgs = tune(T_WGRAD_FUSED_WGS);      // tuning hook: target workgroup count (512 measured best: 2 per CU)
        if (wgs <= 2) wgs = 512;
        split = cdiv(wgs, tiles);
    }
    if (split > units) split = units;
    if (split < 1) split = 1;
    ups = cdiv(units, split);
    return cdiv(units, ups);
}

// f32x3 forward / input gradient on 192 x 128 tiles (conv_x3.h; the rule is in detr_hip_conv3x3_f32)
static void launch_conv_x3_192(const ConvArgs &a0, bool dgrad, hipStream_t s) {
    ConvArgs a = a0;
    a.tiles_m = cdiv(a.M, 192);
    a.tiles_n = cdiv(a.Cd, 128);
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(GEMM_THREADS);
    if (dgrad) hipLaunchKernelGGL((conv3x3_x3_kernel<192, 128, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_x3_kernel<192, 128, false>), grid, block, 0, s, a);
}

// f32x3 forward / input gradient on 128 x 64 tiles, three workgroups per CU
static void launch_conv_x3_128x64(const ConvArgs &a0, bool dgrad, hipStream_t s) {
    ConvArgs a = a0;
    a.tiles_m = cdiv(a.M, 128);
    a.tiles_n = cdiv(a.Cd, 64);
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(GEMM_THREADS);
    if (dgrad) hipLaunchKernelGGL((conv3x3_x3_kernel<128, 64, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_x3_kernel<128, 64, false>), grid, block, 0, s, a);
}

template <int BM, int BN, int WGM, int WGN>
static void launch_wgrad(const ConvWgradArgs &a0, int split, float *ws, long long ws_bytes, hipStream_t s, bool bf16c = false, bool split3 = false) {
    ConvWgradArgs a = a0;
    a.tiles_m = cdiv(a.Ci, BM);
    a.tiles_n = cdiv(a.Co, BN);
    const int tiles = a.tiles_m * a.tiles_n;
    int rps;
    if (split3 && split <= 0 && BM == BN && tune(T_X3_CONV) != 2) split = wgrad_x3_split(a.M, tiles, BM);
    split = wgrad_split_plan(a.M, tiles, split, rps);
    a.rows_per_split = rps;
    const long long part = 9LL * a.Ci * a.Co;
    const bool partial = split > 1 && ws && aligned16(ws) && ws_bytes >= (long long)split * part * 4;
    float *dw_final = a.dw;
    const EpiArgs final_e = a.e;
    a.part_stride = 0;
    a.slab_ts = 0;
    // whole tiles (channel counts are multiples of the tile): the tile-ordered slab has exactly the row-major slab's size
    // (the tile-ordered reduce does 16-byte read-modify-writes of dw and 16-byte slab loads: `partial` already holds aligned16(ws);
    //  a gradient buffer that is only 4-byte aligned keeps the row-major slabs and the scalar reduce -- the rule of gemm_slab_ts)
    const bool ts = partial && WGM == 2 && WGN == 2 && a.Ci % BM == 0 && a.Co % BN == 0 && (!final_e.scale || aligned16(final_e.scale)) &&
                    aligned16(dw_final) && tune(T_SLAB_TS) != 2;
    if (partial) {
        a.dw = ws;
        a.part_stride = part;
        a.e.alpha = 1.0f;
        a.e.scale = nullptr;
        a.e.atomic = 0;
        a.e.vec = 1;
        a.slab_ts = ts ? 1 : 0;
    } else if (split == 1) {
        a.e.atomic = 1;   // accumulate onto dw
    }
    dim3 grid((unsigned)tiles, 9, (unsigned)split), block(GEMM_THREADS);
    if (bf16c && a.s16) hipLaunchKernelGGL((conv3x3_wgrad_bf16c_kernel<BM, BN, WGM, WGN, true>), grid, block, 0, s, a);
    else if (bf16c) hipLaunchKernelGGL((conv3x3_wgrad_bf16c_kernel<BM, BN, WGM, WGN, false>), grid, block, 0, s, a);
    else if (split3 && BM == BN && WGM == 2 && WGN == 2 && tune(T_X3_CONV) != 2) {
        if constexpr (BM == BN && WGM == 2 && WGN == 2) hipLaunchKernelGGL((conv3x3_wgrad_x3_kernel<BM, BN>), grid, block, 0, s, a);
    } else if (split3) hipLaunchKernelGGL((conv3x3_wgrad_kernel<BM, BN, WGM, WGN, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_wgrad_kernel<BM, BN, WGM, WGN>), grid, block, 0, s, a);
    if (partial) launch_splitk_reduce(ws, split, part, 9 * a.Ci, a.Co, dw_final, a.Co, final_e.alpha, final_e.scale, s, nullptr, nullptr, 1.0f,
                                      ts ? BM : 0, ts ? BN : 0, a.tiles_n);
}

// fused-tap bf16 weight gradient (stride 1, pad 1, channel counts % 64 == 0)
static void launch_wgrad_fused(const ConvWgradArgs &a0, int split, float *ws, long long ws_bytes, hipStream_t s) {
    ConvWgradArgs a = a0;
    a.tiles_m = a.Ci / 64;
    a.tiles_n = a.Co / 64;
    const int tiles = a.tiles_m * a.tiles_n;
    const int chunks = cdiv(a.Wo, 32);
    const int units = a.N * a.Ho * chunks;
    int ups;
    split = wgrad_fused_split_plan(units, tiles, split, ups);
    const long long part = 9LL * a.Ci * a.Co;
    const bool partial = split > 1 && ws && aligned16(ws) && ws_bytes >= (long long)split * part * 4;
    float *dw_final = a.dw;
    const EpiArgs final_e = a.e;
    a.part_stride = 0;
    a.slab_ts = 0;
    const bool ts = partial && (!final_e.scale || aligned16(final_e.scale)) && aligned16(dw_final) && tune(T_SLAB_TS) != 2;
    if (partial) {
        a.dw = ws;
        a.part_stride = part;
        a.e.alpha = 1.0f;
        a.e.scale = nullptr;
        a.e.atomic = 0;
        a.e.vec = 1;
        a.slab_ts = ts ? 1 : 0;
    } else {
        a.e.atomic = 1;   // accumulate onto dw (split == 1, or the atomic fallback without a workspace)
    }
    dim3 grid((unsigned)tiles, 1, (unsigned)split), block(GEMM_THREADS);
    if (a.stride == 2) hipLaunchKernelGGL((conv3x3_wgrad_fused_bf16_kernel<true, 2>), grid, block, 0, s, a, ups, chunks);
    else if (a.s16) hipLaunchKernelGGL(conv3x3_wgrad_fused_bf16_kernel<true>, grid, block, 0, s, a, ups, chunks);
    else hipLaunchKernelGGL(conv3x3_wgrad_fused_bf16_kernel<false>, grid, block, 0, s, a, ups, chunks);
    if (partial) launch_splitk_reduce(ws, split, part, 9 * a.Ci, a.Co, dw_final, a.Co, final_e.alpha, final_e.scale, s, nullptr, nullptr, 1.0f,
                                      ts ? 64 : 0, ts ? 64 : 0, a.tiles_n);
}

}  // namespace detr

using namespace detr;

static bool wgrad_is_fused(const detr_conv3x3_desc *d) {
    const bool bf = d->compute == 1 && d->Ci % 32 == 0 && d->Co % 32 == 0;
    // stride 2 (round 5): bf16-stored tensors only, and only up to 128 channels -- measured (scripts/micro_conv.py, profiles/r05_micro_conv_dma.txt;
    // nine-tap | per-tap, us): 200x334x128 -> 100x167 100.1 | 115.9, 100x167x256 -> 50x84 96.7 | 90.5, 50x84x512 -> 25x42 116.0 | 105.0: the stride-2
    // patch is twice the bytes per MFMA of the stride-1 one and every (ci, co) tile re-fetches it.  DETR_HIP_WGRAD_FUSED = 3: per-tap kernel for all
    // stride-2 convolutions, 4: nine-tap kernel for all (A/B).
    const int fmode = tune(T_WGRAD_FUSED);
    const bool s_ok = d->stride == 1 || (d->stride == 2 && d->x_dtype == 1 && d->w_dtype == 1 && fmode != 3 && (fmode == 4 || (d->Ci <= 128 && d->Co <= 128)));
    return bf && s_ok && d->pad == 1 && d->Ci % 64 == 0 && d->Co % 64 == 0 && tune(T_WGRAD_FUSED) != 2;
}

extern "C" int64_t detr_hip_workspace_bytes_conv3x3(const detr_conv3x3_desc *d, int32_t mode) {
    if (!d || d->N <= 0 || d->Ci <= 0 || d->Co <= 0) return -1;
    if (mode != 2) return 0;                       // forward / input gradient: no scratch
    const long long part = 9LL * d->Ci * d->Co;
    int split, aux;
    if (wgrad_is_fused(d)) {
        split = wgrad_fused_split_plan(d->N * d->Ho * cdiv(d->Wo, 32), (d->Ci / 64) * (d->Co / 64), d->split, aux);
    } else {
        const bool bf = d->compute == 1 && d->Ci % 32 == 0 && d->Co % 32 == 0;
        const int wforce = bf ? 0 : tune(T_WGRAD_TILE);
        const int t = (wforce == 3) ? 64 : ((wforce == 1 || (d->Ci >= 128 && d->Co >= 128)) ? 128 : 64);
        int ask = d->split;
        if (d->compute == 2 && ask <= 0 && tune(T_X3_CONV) != 2) ask = wgrad_x3_split(d->N * d->Ho * d->Wo, cdiv(d->Ci, t) * cdiv(d->Co, t), t);      // (the launch's plan)
        split = wgrad_split_plan(d->N * d->Ho * d->Wo, cdiv(d->Ci, t) * cdiv(d->Co, t), ask, aux);
    }
    return split > 1 ? (int64_t)split * part * 4 : 0;
}

extern "C" int detr_hip_conv3x3_f32(const detr_conv3x3_desc *d, int32_t mode, void *stream) {
    DETR_REQUIRE(d != nullptr, "conv3x3: null descriptor");
    DETR_REQUIRE(mode >= 0 && mode <= 2, "conv3x3: bad mode %d", mode);
    DETR_REQUIRE(d->stride == 1 || d->stride == 2, "conv3x3: stride %d unsupported", d->stride);
    DETR_REQUIRE(d->Ci % 16 == 0 && d->Co % 16 == 0, "conv3x3: Ci=%d Co=%d must be multiples of 16", d->Ci, d->Co);
    DETR_REQUIRE(d->Ho == (d->Hi + 2 * d->pad - 3) / d->stride + 1 && d->Wo == (d->Wi + 2 * d->pad - 3) / d->stride + 1,
                 "conv3x3: output %dx%d inconsistent with input %dx%d pad %d stride %d", d->Ho, d->Wo, d->Hi, d->Wi,
                 d->pad, d->stride);
    DETR_REQUIRE(d->x && d->w && d->y, "conv3x3: null operand");
    DETR_REQUIRE(aligned16(d->x) && aligned16(d->w) && aligned16(d->y), "conv3x3: operands must be 16-byte aligned");
    DETR_REQUIRE((long long)d->N * d->Hi * d->Wi * d->Ci * 4 <= BUF_MAX_BYTES &&
                     (long long)d->N * d->Ho * d->Wo * d->Co * 4 <= BUF_MAX_BYTES,
                 "conv3x3: a tensor spans more than 4 GB (32-bit buffer offsets)");
    hipStream_t s = (hipStream_t)stream;
    EpiArgs e;
    e.alpha = d->alpha;
    e.scale = d->scale;
    e.bias = d->bias;
    e.residual = d->residual;
    e.mask = d->mask;
    e.act = d->act;
    e.atomic = 0;
    e.drop_scale = 0.0f; e.drop_thresh = 0; e.drop_seed = 0;
    const bool bits_in = d->mask && d->m_dtype == 2, bits_out = mode == 0 && d->maskbits_out != nullptr;
    e.vec = (!d->scale || aligned16(d->scale)) && (!d->bias || aligned16(d->bias)) &&
            (!d->residual || aligned16(d->residual)) && (!d->mask || bits_in || aligned16(d->mask));   // channel counts are % 16
    e.c16 = (mode != 2 && d->y_dtype == 1); e.r16 = d->r_dtype == 1;
    e.m16 = (d->m_dtype == 1 || d->m_dtype == 2) ? d->m_dtype : 0;          // 2: bit-packed mask (see detr_gemm_desc.maskbits_out)
    if (mode == 2) {
        DETR_REQUIRE(!d->bias && !d->residual && !d->mask && d->act == 0, "conv3x3 wgrad: only scale/alpha epilogue");
        ConvWgradArgs a;
        a.N = d->N; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
        a.stride = d->stride; a.pad = d->pad;
        a.x = d->x; a.dy = d->w; a.dw = d->y;
        a.M = d->N * d->Ho * d->Wo;
        a.s16 = (d->x_dtype == 1);
        DETR_REQUIRE(d->x_dtype == d->w_dtype && d->y_dtype == 0, "conv3x3 wgrad: x and dy must share their storage type, dw is fp32");
        DETR_REQUIRE(!a.s16 || (d->compute == 1 && d->Ci % 32 == 0 && d->Co % 32 == 0), "conv3x3 wgrad: bf16 tensors need compute = bf16");
        e.atomic = 1; e.ldr = 0; e.ldmask = 0;
        a.e = e;
        const bool bf = d->compute == 1 && d->Ci % 32 == 0 && d->Co % 32 == 0;
        if (wgrad_is_fused(d)) {
            launch_wgrad_fused(a, d->split, d->workspace, d->workspace_bytes, s);
            DETR_LAUNCH_CHECK("conv3x3 wgrad bf16 (fused taps)");
            return 0;
        }
        if (bf) {
            if (d->Ci >= 128 && d->Co >= 128) launch_wgrad<128, 128, 2, 2>(a, d->split, d->workspace, d->workspace_bytes, s, true);
            else launch_wgrad<64, 64, 2, 2>(a, d->split, d->workspace, d->workspace_bytes, s, true);
            DETR_LAUNCH_CHECK("conv3x3 wgrad bf16");
            return 0;
        }
        const int wforce = tune(T_WGRAD_TILE);
        const bool s3 = d->compute == 2;
        if (wforce == 3) launch_wgrad<64, 64, 2, 2>(a, d->split, d->workspace, d->workspace_bytes, s, false, s3);
        else if (wforce == 1) launch_wgrad<128, 128, 2, 2>(a, d->split, d->workspace, d->workspace_bytes, s, false, s3);
        else if (d->Ci >= 128 && d->Co >= 128) launch_wgrad<128, 128, 2, 2>(a, d->split, d->workspace, d->workspace_bytes, s, false, s3);
        else launch_wgrad<64, 64, 2, 2>(a, d->split, d->workspace, d->workspace_bytes, s, false, s3);
        DETR_LAUNCH_CHECK("conv3x3 wgrad");
        return 0;
    }
    ConvArgs a;
    a.N = d->N; a.stride = d->stride; a.pad = d->pad; a.Ci = d->Ci; a.Co = d->Co;
    a.w = d->w; a.src = d->x; a.dst = d->y;
    if (mode == 0) {
        a.Hs = d->Hi; a.Ws = d->Wi; a.Cs = d->Ci;
        a.Hd = d->Ho; a.Wd = d->Wo; a.Cd = d->Co;
    } else {
        a.Hs = d->Ho; a.Ws = d->Wo; a.Cs = d->Co;
        a.Hd = d->Hi; a.Wd = d->Wi; a.Cd = d->Ci;
    }
    a.M = d->N * a.Hd * a.Wd;
    e.ldr = a.Cd;
    e.ldmask = bits_in ? a.Cd / 8 : a.Cd;
    e.mbits_out = bits_out ? d->maskbits_out : nullptr;
    e.ld_mbits_out = a.Cd / 8;
    // all-bf16 epilogue streams (channel counts are % 16, rows 16-byte aligned): 8 columns per lane (gemm_core.h epilogue_wide16)
    e.wide16 = e.c16 && e.vec && (!d->residual || e.r16) && (!d->mask || e.m16) && (tune(T_EPI_WIDE) != 2 || bits_in || bits_out);
    DETR_REQUIRE(!(bits_in || bits_out) || (d->compute == 1 && e.wide16), "conv3x3: bit-packed masks need bf16 x / y / residual tensors (compute = bf16)");
    a.e = e;
    a.w16 = (d->w_dtype == 1);
    a.x16 = (d->x_dtype == 1);
    DETR_REQUIRE((d->x_dtype == 0 && d->y_dtype == 0 && d->r_dtype == 0 && d->m_dtype == 0) || (d->compute == 1 && d->Ci % 32 == 0 && d->Co % 32 == 0),
                 "conv3x3: bf16 tensors need compute = bf16 and channel counts %% 32 == 0");
    DETR_REQUIRE(d->w_dtype == 0 || (d->w_dtype == 1 && d->compute == 1 && mode != 2 && d->Ci % 32 == 0 && d->Co % 32 == 0),
                 "conv3x3: a bf16 kernel tensor needs compute = bf16, mode 0/1 and channel counts %% 32 == 0");
    a.par_on = 0; a.Hp = a.Wp = a.ph = a.pw = a.kh0 = a.kw0 = 0; a.nth = a.ntw = 3;
    a.cls_mix = 0;
    const bool dgrad = mode == 1;
    // stride-1 convs on bf16 tensors: the halo-staged kernel (conv_halo.h; DETR_HIP_CONV_HALO=2 = off), for 128-channel panels on 4-row tiles
    // its LDS-DMA form (conv_halo_dma.h: bit-identical, 5-10 % faster; DETR_HIP_CONV_DMA=2 = off).  With 4-row tiles it also
    // takes the 512-channel convs of layer4 (25 x 42 maps: 92.9 -> 64.5 us forward, 103.2 -> 65.5 us input gradient against the
    // 64x64 implicit-GEMM tiles; with 8-row tiles -- 64 workgroups x 4 channel slices, half of every tile padding -- it lost).
    if (d->compute == 1 && d->stride == 1 && d->pad == 1 && a.w16 && a.x16 && e.c16 && a.Cs % 32 == 0 && a.Cd % 64 == 0 &&
        a.Cd <= 512 && (!d->mask || e.m16) && !d->residual && !d->scale && d->alpha == 1.0f && (d->act == 0 || d->act == 1) &&
        tune(T_CONV_HALO) != 2) {
        if (a.Cd >= 128 && a.Cd % 128 == 0 && tune(T_CONV_DMA) != 2 && tune(T_CONV_HALO) != 4) { if (launch_conv_halo_dma(a, dgrad, s)) return -1; }
        else if (a.Cd >= 128) { if (launch_conv_halo<128>(a, dgrad, s)) return -1; }
        else if (launch_conv_halo<64>(a, dgrad, s)) return -1;
        DETR_LAUNCH_CHECK("conv3x3 (halo)");
        return 0;
    }
    const int force = tune(T_CONV_TILE);     // tuning hook; 0 = heuristic
    auto launch = [&](const ConvArgs &c) {
        const long long big = (long long)cdiv(c.M, 128) * cdiv(c.Cd, 128);
        if (d->compute == 1 && c.Cs % 32 == 0 && c.Cd % 32 == 0) {
            // measured: profiles/tune_bf16_r1d.txt; round 3 (scripts/micro_conv.py, with the wide epilogue): the stride-2 forward at 128
            // channels 83 -> 72 us (cold 113 -> 97) and its input-gradient classes 173 -> 165 us on 128x128 tiles
            if (force == 3 || (force == 0 && (c.Cd < 128 || big < 256)))
                launch_conv_bf16<64, 64, 2, 2>(c, dgrad, s);
            else launch_conv_bf16<128, 128, 2, 2>(c, dgrad, s);
        } else if (d->compute == 2) {
            // f32x3: 64 x 64 wave tiles (128 x 128 workgroup tiles) are matrix-pipe bound, 32 x 32 ones VALU bound (see gemm_pick_tile)
            const int lim = tune(T_SPLIT3_T128) > 0 ? tune(T_SPLIT3_T128) : 192;
            // 192 x 128 tiles where 128 x 128 ones overflow the 512 workgroup slots (two per CU) by a few: 256 channels at 50 x 84 are 526 tiles = two
            // rounds, 350 tiles of 192 rows one round of 1.5 x the work (the rule of gemm_pick_tile's tile 8; DETR_HIP_X3_T192 = 2: never, 1: always)
            const long long t192 = (long long)cdiv(c.M, 192) * cdiv(c.Cd, 128);
            const long long r128 = (big + 511) / 512, r192 = (t192 + 511) / 512;
            const bool x3form = c.Cs % 32 == 0 && tune(T_X3_CONV) != 2 && (!c.par_on || tune(T_X3_CONV) == 1);
            if (force == 3 || (force == 0 && (c.Cd < 128 || big < lim))) launch_conv<64, 64, 2, 2>(c, dgrad, s, true);
            else if (force == 0 && x3form && tune(T_X3_T192) != 2 && tune(T_X3_T192) != 3 && (tune(T_X3_T192) == 1 || 3 * r192 < 2 * r128)) launch_conv_x3_192(c, dgrad, s);
            // 128 x 64 tiles, three workgroups per CU: one 128-channel panel (layer2: 1044 tiles of 128 x 128 = 2.04 rounds -> 290 vs 275 us) and grids
            // that leave CUs with a single 128 x 128 workgroup (layer4: 264 tiles -> 357 vs 338 us); DETR_HIP_X3_T192 = 3: wherever 128 x 128 would run
            else if (force == 0 && x3form && tune(T_X3_T192) != 2 && (tune(T_X3_T192) == 3 || c.Cd == 128 || big <= 384)) launch_conv_x3_128x64(c, dgrad, s);
            else launch_conv<128, 128, 2, 2>(c, dgrad, s, true);
        } else if (force == 1) launch_conv<128, 128, 2, 2>(c, dgrad, s);
        else if (force == 2) launch_conv<128, 64, 2, 2>(c, dgrad, s);
        else launch_conv<64, 64, 2, 2>(c, dgrad, s);   // 64x64 measured best on every backbone shape (profiles/tune_r1.txt)
    };
    if (dgrad && d->stride == 2 && tune(T_DGRAD_S2_CLASSES) != 2) {
        // one launch per destination-pixel parity class: 1 + 2 + 2 + 4 tap-GEMMs instead of 9 with 3/4 of the rows masked
        // all-bf16 tensors: the class form of the halo-staged kernel (conv_halo.h; DETR_HIP_DGRAD_S2_CLASSES = 3: tile kernel)
        const bool halo_cls = d->compute == 1 && d->pad == 1 && a.w16 && a.x16 && e.c16 && a.Cs % 32 == 0 && a.Cd % 64 == 0 &&
                              a.Cd <= 512 && (!d->mask || e.m16) && !d->residual && !d->scale && !d->bias && d->alpha == 1.0f &&
                              d->act == 0 && tune(T_DGRAD_S2_CLASSES) != 3 && tune(T_CONV_HALO) != 2;
        if (halo_cls) {
            if (a.Cd >= 128) { if (launch_conv_halo_s2classes<128>(a, s)) return -1; }
            else if (launch_conv_halo_s2classes<64>(a, s)) return -1;
            DETR_LAUNCH_CHECK("conv3x3 (halo, stride-2 classes)");
            return 0;
        }
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                ConvArgs c = a;
                c.par_on = 1; c.ph = ph; c.pw = pw;

                c.Hp = (a.Hd + 1 - ph) / 2;
                c.Wp = (a.Wd + 1 - pw) / 2;
                c.M = d->N * c.Hp * c.Wp;
                if (c.M <= 0) continue;
                c.kh0 = (ph + d->pad) & 1; c.kw0 = (pw + d->pad) & 1;
                c.nth = c.kh0 ? 1 : 2; c.ntw = c.kw0 ? 1 : 2;
                c.e.remap_w2 = c.Wp; c.e.remap_h2 = c.Hp; c.e.remap_W = a.Wd; c.e.remap_H = a.Hd;
                c.e.remap_ph = ph; c.e.remap_pw = pw;
                launch(c);
            }
    } else {
        launch(a);
    }
    DETR_LAUNCH_CHECK("conv3x3");
    return 0;
}

extern "C" int detr_hip_maxpool3x3s2_fwd_f32(const float *x, float *y, uint8_t *argmax, int32_t N, int32_t H, int32_t W,
                                             int32_t C, int32_t Ho, int32_t Wo, void *stream) {
    DETR_REQUIRE(x && y && argmax, "maxpool fwd: null operand");
    DETR_REQUIRE(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "maxpool fwd: bad output size");
    const long long total = (long long)N * Ho * Wo * C;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, N,
                       H, W, C, Ho, Wo, total);
    DETR_LAUNCH_CHECK("maxpool fwd");
    return 0;
}

extern "C" int detr_hip_maxpool3x3s2_bwd_f32(const float *dy, const uint8_t *argmax, const float *x, float *dx, int32_t N,
                                             int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, void *stream) {
    DETR_REQUIRE(dy && argmax && x && dx, "maxpool bwd: null operand");
    DETR_REQUIRE(C % 4 == 0 && aligned16(dy) && aligned16(x) && aligned16(dx) && ((uintptr_t)argmax % 4 == 0),
                 "maxpool bwd: C must be a multiple of 4 and the tensors 16-byte aligned");
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, argmax, x,
                       dx, N, H, W, C, Ho, Wo, total);
    DETR_LAUNCH_CHECK("maxpool bwd");
    return 0;
}

extern "C" int detr_hip_subsample2_fwd_f32(const float *x, float *y, int32_t N, int32_t H, int32_t W, int32_t C,
                                           int32_t Ho, int32_t Wo, void *stream) {
    DETR_REQUIRE(x && y && C % 4 == 0 && aligned16(x) && aligned16(y), "subsample2 fwd: bad operands");
    DETR_REQUIRE(Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "subsample2 fwd: bad output size");
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(subsample2_fwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)x, (float4 *)y, H, W, C / 4, Ho, Wo, total);
    DETR_LAUNCH_CHECK("subsample2 fwd");
    return 0;
}

extern "C" int detr_hip_subsample2_bwd_f32(const float *dy, float *dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                           int32_t Ho, int32_t Wo, void *stream) {
    DETR_REQUIRE(dy && dx && C % 4 == 0 && aligned16(dy) && aligned16(dx), "subsample2 bwd: bad operands");
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)dy, (float4 *)dx, H, W, C / 4, Ho, Wo, total);
    DETR_LAUNCH_CHECK("subsample2 bwd");
    return 0;
}

extern "C" int detr_hip_maxpool3x3s2_fwd_bf16(const uint16_t *x, uint16_t *y, uint8_t *argmax, int32_t N, int32_t H, int32_t W,
                                              int32_t C, int32_t Ho, int32_t Wo, void *stream) {
    DETR_REQUIRE(x && y && argmax && C % 4 == 0, "maxpool fwd bf16: bad operands");
    DETR_REQUIRE(((uintptr_t)x % 8 == 0) && ((uintptr_t)y % 8 == 0) && ((uintptr_t)argmax % 4 == 0), "maxpool fwd bf16: alignment");
    DETR_REQUIRE(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "maxpool fwd: bad output size");
    if (C % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)argmax % 8 == 0)) {
        const long long total8 = (long long)N * Ho * Wo * (C / 8);
        hipLaunchKernelGGL(maxpool_fwd_bf16x8_kernel, dim3(ew_grid(total8, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)x,
                           (uint4 *)y, (uint2 *)argmax, N, H, W, C / 8, Ho, Wo, total8);
        DETR_LAUNCH_CHECK("maxpool fwd bf16 (8 channels per lane)");
        return 0;
    }
    const long long total = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_bf16_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint2 *)x,
                       (uint2 *)y, (uint32_t *)argmax, N, H, W, C / 4, Ho, Wo, total);
    DETR_LAUNCH_CHECK("maxpool fwd bf16");
    return 0;
}

extern "C" int detr_hip_maxpool3x3s2_bwd_bf16(const uint16_t *dy, const uint8_t *argmax, const uint16_t *x, uint16_t *dx, int32_t N,
                                              int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, void *stream) {
    DETR_REQUIRE(dy && argmax && x && dx && C % 4 == 0, "maxpool bwd bf16: bad operands");
    DETR_REQUIRE(((uintptr_t)dy % 8 == 0) && ((uintptr_t)x % 8 == 0) && ((uintptr_t)dx % 8 == 0) && ((uintptr_t)argmax % 4 == 0),
                 "maxpool bwd bf16: alignment");
    if (C % 8 == 0 && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dx % 16 == 0) && ((uintptr_t)argmax % 8 == 0)) {
        const long long total8 = (long long)N * H * W * (C / 8);
        hipLaunchKernelGGL(maxpool_bwd_bf16x8_kernel, dim3(ew_grid(total8, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)dy,
                           (const uint2 *)argmax, (const uint4 *)x, (uint4 *)dx, N, H, W, C / 8, Ho, Wo, total8);
        DETR_LAUNCH_CHECK("maxpool bwd bf16 (8 channels per lane)");
        return 0;
    }
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint2 *)dy,
                       (const uint32_t *)argmax, (const uint2 *)x, (uint2 *)dx, N, H, W, C / 4, Ho, Wo, total);
    DETR_LAUNCH_CHECK("maxpool bwd bf16");
    return 0;
}
