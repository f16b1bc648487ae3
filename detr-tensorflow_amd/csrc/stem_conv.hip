// stem_conv.hip -- the ResNet stem convolution (ZeroPadding2D(3) + 7x7 stride-2 VALID conv 3 -> 64 + frozen BN + ReLU,
// detr_tf/networks/resnet_backbone.py:11-26) as an IMPLICIT GEMM, forward and weight gradient, for gfx950.
//
// rows = output pixels (n, ho, wo), K = 147 = (kh, kw, c) in HWIO order, N = 64 output channels.  The A operand is never
// materialised: the loader gathers img[n, 2*ho-3+kh, 2*wo-3+kw, c] straight from the NHWC image through a buffer descriptor
// (out-of-image taps and k >= 147 take the out-of-range offset -> zeros).  With C = 3 the 21 floats (kw, c) of one kernel row
// are contiguous in the image, so k -> address is  (n*H + hi)*3W + (2*wo-3)*3 + (k - 21*kh).  The image (102 MB at B = 8,
// 800x1333) stays L2 / MALL resident while each element is used by ~12 output pixels; the former im2col buffer
// (2.13 M x 160 floats = 1.37 GB written once and read twice per step) is gone.
// Tile engine, epilogue and LDS images are the ones of gemm_core.h / gemm_bf16_core.h (bf16 mode: the gathered tile
// of the weight gradient and dy are staged as transpose-read images).
#include "gemm_core.h"
#include "gemm_bf16_core.h"

namespace detr {

constexpr int STEM_K = 147;

struct StemArgs {
    int N, H, W, Ho, Wo, M;      // M = N*Ho*Wo output pixels
    const float *img;            // [N, H, W, 3]
    const float *w;              // fwd: [147][64] (HWIO flattened, BN scale folded by the caller)
    float *y;                    // fwd: [M, 64]
    const float *dy;             // wgrad: [M, 64]
    float *dw;                   // wgrad: [147][64] (or partial slabs)
    int rows_per_split;
    long long part_stride;
    EpiArgs e;
};

struct StemRow {                 // per output pixel: element offset of its image, first input row / flat column
    unsigned imgbase;
    int hbase, fcbase;
    bool ok;
    __device__ __forceinline__ void set(const StemArgs &a, int m) {
        ok = m < a.M;
        const int mm = ok ? m : 0;
        const int wo = mm % a.Wo;
        const int t = mm / a.Wo;
        const int ho = t % a.Ho;
        const int n = t / a.Ho;
        imgbase = (unsigned)n * (unsigned)(a.H * a.W * 3);
        hbase = 2 * ho - 3;
        fcbase = (2 * wo - 3) * 3;
    }
};

__device__ __forceinline__ float stem_gather1(const BufSrc &src, const StemArgs &a, const StemRow &r, int k) {
    const int kh = k / 21;
    const int hi = r.hbase + kh;
    const int fc = r.fcbase + k - 21 * kh;
    const bool v = r.ok && k < STEM_K && hi >= 0 && hi < a.H && fc >= 0 && fc < 3 * a.W;
    return src.ld1(v ? (r.imgbase + (unsigned)(hi * a.W * 3 + fc)) * 4u : BUF_OOB);
}
__device__ __forceinline__ float4 stem_gather4(const BufSrc &src, const StemArgs &a, const StemRow &r, int k) {
    float4 v;
    v.x = stem_gather1(src, a, r, k);
    v.y = stem_gather1(src, a, r, k + 1);
    v.z = stem_gather1(src, a, r, k + 2);
    v.w = stem_gather1(src, a, r, k + 3);
    return v;
}

// ------------------------------------------------------------------------------------------------
// forward: y = act((gather(img) @ w) + bias)      tile 64 pixels x 64 channels
// ------------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(GEMM_THREADS) void stem_fwd_kernel(StemArgs a) {
    constexpr int BM = 64, BN = 64, BK = BF ? BF_BK : GEMM_BK;
    constexpr int SMEM = BF ? BfSmemBytes<BM, BN, 2>::VALUE : SmemBytes<BM, BN, 2>::VALUE;
    __shared__ __attribute__((aligned(16))) char smem_raw[SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    BufSrc src;
    src.init(a.img, (long long)a.N * a.H * a.W * 3);
    constexpr int NVA = BF ? 2 : 1;                  // A float4 per thread and K tile (LoaderKb / LoaderK thread maps)
    StemRow rows[NVA];
    const int kq = BF ? (tid & 7) * 4 : (tid & 3) * 4;
#pragma unroll
    for (int i = 0; i < NVA; ++i) rows[i].set(a, m0 + (BF ? (tid >> 3) + 32 * i : (tid >> 2)));

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
    constexpr int nkt = (STEM_K + BK - 1) / BK;
    float4 ra[NVA];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) ra[i] = stem_gather4(src, a, rows[i], kt * BK + kq);
    };
    if constexpr (BF) {
        BfSmem<BM, BN> &sm = *reinterpret_cast<BfSmem<BM, BN> *>(smem_raw);
        LoaderMNt<BN> lb;
        lb.init(a.w, 64, 0, 64, STEM_K, true, tid);
        float4 rb[LoaderMNt<BN>::NU];
        auto store_a = [&](unsigned short (*S)[BF_LD]) {
#pragma unroll
            for (int i = 0; i < NVA; ++i)
                *reinterpret_cast<uint2 *>(&S[(tid >> 3) + 32 * i][kq]) = make_uint2(pack_bf16(ra[i].x, ra[i].y), pack_bf16(ra[i].z, ra[i].w));
        };
        load_a(0);
        lb.load(0, STEM_K, rb);
        store_a(sm.A[0]);
        lb.store(sm.B[0], rb);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            const bool more = (kt + 1) < nkt;
            if (more) {
                load_a(kt + 1);
                lb.load((kt + 1) * BK, STEM_K, rb);
            }
            mma_ktile_bf16<BM, BN, 2, 2, false, true>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
            if (more) {
                store_a(sm.A[cur ^ 1]);
                lb.store(sm.B[cur ^ 1], rb);
            }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        GemmSmem<BM, BN> &sm = *reinterpret_cast<GemmSmem<BM, BN> *>(smem_raw);
        LoaderMN<BN> lb;
        lb.init(a.w, 64, 0, 64, STEM_K, true, tid);
        float4 rb[LoaderMN<BN>::NV];
        auto store_a = [&](float (*S)[GemmSmem<BM, BN>::LDA]) {
            const int row = tid >> 2;
            S[kq + 0][row] = ra[0].x; S[kq + 1][row] = ra[0].y; S[kq + 2][row] = ra[0].z; S[kq + 3][row] = ra[0].w;
        };
        load_a(0);
        lb.load(0, STEM_K, rb);
        store_a(sm.A[0]);
        lb.store(sm.B[0], rb);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            const bool more = (kt + 1) < nkt;
            if (more) {
                load_a(kt + 1);
                lb.load((kt + 1) * BK, STEM_K, rb);
            }
            mma_ktile<BM, BN, 2, 2>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
            if (more) {
                store_a(sm.A[cur ^ 1]);
                lb.store(sm.B[cur ^ 1], rb);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    epilogue<BM, BN, 2, 2>(acc, reinterpret_cast<float *>(smem_raw), a.y, 64, a.M, 64, m0, 0, wm, wn, lane, wave, a.e);
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dw[k][co] (+)= sum_m gather(img)[m][k] * dy[m][co]
// grid = (3 tiles of 64 k, 1, row splits); reduction over the output pixels of the split
// ------------------------------------------------------------------------------------------------
template <bool BF, bool D16>
__global__ __launch_bounds__(GEMM_THREADS) void stem_wgrad_kernel(StemArgs a) {
    constexpr int BM = 64, BN = 64, BK = BF ? BF_BK : GEMM_BK;
    constexpr int SMEM = BF ? BfSmemBytes<BM, BN, 2>::VALUE : SmemBytes<BM, BN, 2>::VALUE;
    __shared__ __attribute__((aligned(16))) char smem_raw[SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int k0 = blockIdx.x * BM;                       // first weight row (k index) of this tile
    const int m_begin = blockIdx.z * a.rows_per_split;
    const int m_end = min(a.M, m_begin + a.rows_per_split);
    if (m_begin >= m_end) return;
    const int nkt = (m_end - m_begin + BK - 1) / BK;
    BufSrc src;
    src.init(a.img, (long long)a.N * a.H * a.W * 3);

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
    // A' tile [BK reduction rows (pixels)][64 k]: thread maps of LoaderMN (fp32) / LoaderMNt (bf16)
    constexpr int NUA = BF ? 2 : 1;
    float4 ra[NUA];
    auto load_a = [&](int mt0) {
#pragma unroll
        for (int i = 0; i < NUA; ++i) {
            int mrow, kcol;
            if (BF) {
                const int u = tid + 256 * i;
                mrow = 4 * (u >> 6) + ((u >> 2) & 3);
                kcol = 16 * ((u >> 4) & 3) + 4 * (u & 3);
            } else {
                mrow = tid >> 4;
                kcol = (tid & 15) * 4;
            }
            StemRow r;
            const int m = mt0 + mrow;
            r.set(a, m);
            r.ok = r.ok && m < m_end;
            ra[i] = stem_gather4(src, a, r, k0 + kcol);
        }
    };
    if constexpr (BF) {
        BfSmem<BM, BN> &sm = *reinterpret_cast<BfSmem<BM, BN> *>(smem_raw);
        typename std::conditional<D16, LoaderMNth<BN>, LoaderMNt<BN>>::type lb;     // D16: dy is bf16 in memory
        lb.init(a.dy, 64, 0, 64, a.M, true, tid);
        typename std::conditional<D16, uint2, float4>::type rb[LoaderMNt<BN>::NU];
        auto store_a = [&](unsigned short (*S)[BF_LD]) {
            unsigned short *flat = &S[0][0];
#pragma unroll
            for (int i = 0; i < NUA; ++i)
                *reinterpret_cast<uint2 *>(flat + (tid + 256 * i) * 4) = make_uint2(pack_bf16(ra[i].x, ra[i].y), pack_bf16(ra[i].z, ra[i].w));
        };
        load_a(m_begin);
        lb.load(m_begin, m_end, rb);
        store_a(sm.A[0]);
        lb.store(sm.B[0], rb);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            const bool more = (kt + 1) < nkt;
            if (more) {
                load_a(m_begin + (kt + 1) * BK);
                lb.load(m_begin + (kt + 1) * BK, m_end, rb);
            }
            mma_ktile_bf16<BM, BN, 2, 2, true, true>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
            if (more) {
                store_a(sm.A[cur ^ 1]);
                lb.store(sm.B[cur ^ 1], rb);
            }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        GemmSmem<BM, BN> &sm = *reinterpret_cast<GemmSmem<BM, BN> *>(smem_raw);
        LoaderMN<BN> lb;
        lb.init(a.dy, 64, 0, 64, a.M, true, tid);
        float4 rb[LoaderMN<BN>::NV];
        auto store_a = [&](float (*S)[GemmSmem<BM, BN>::LDA]) {
            *reinterpret_cast<float4 *>(&S[tid >> 4][(tid & 15) * 4]) = ra[0];
        };
        load_a(m_begin);
        lb.load(m_begin, m_end, rb);
        store_a(sm.A[0]);
        lb.store(sm.B[0], rb);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            const bool more = (kt + 1) < nkt;
            if (more) {
                load_a(m_begin + (kt + 1) * BK);
                lb.load(m_begin + (kt + 1) * BK, m_end, rb);
            }
            mma_ktile<BM, BN, 2, 2>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
            if (more) {
                store_a(sm.A[cur ^ 1]);
                lb.store(sm.B[cur ^ 1], rb);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    float *dw = a.dw + (long long)blockIdx.z * a.part_stride;
    epilogue<BM, BN, 2, 2>(acc, reinterpret_cast<float *>(smem_raw), dw, 64, STEM_K, 64, k0, 0, wm, wn, lane, wave, a.e);
}

}  // namespace detr

using namespace detr;

extern "C" int detr_hip_stem_conv7x7_f32(const detr_stem_desc *d, int32_t mode, void *stream) {
    DETR_REQUIRE(d != nullptr, "stem conv: null descriptor");
    DETR_REQUIRE(mode == 0 || mode == 2, "stem conv: mode %d (0 = forward, 2 = weight gradient)", mode);
    DETR_REQUIRE(d->img && d->w && d->y, "stem conv: null operand");
    DETR_REQUIRE(d->Ho == (d->H + 6 - 7) / 2 + 1 && d->Wo == (d->W + 6 - 7) / 2 + 1, "stem conv: bad output size");
    DETR_REQUIRE(aligned16(d->w) && aligned16(d->y) && ((uintptr_t)d->img % 4 == 0), "stem conv: alignment");
    DETR_REQUIRE((long long)d->N * d->H * d->W * 3 * 4 <= BUF_MAX_BYTES && (long long)d->N * d->Ho * d->Wo * 64 * 4 <= BUF_MAX_BYTES,
                 "stem conv: a tensor spans more than 4 GB (32-bit buffer offsets)");
    hipStream_t s = (hipStream_t)stream;
    StemArgs a = {};
    a.N = d->N; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo;
    a.M = d->N * d->Ho * d->Wo;
    a.img = d->img;
    EpiArgs e;
    e.alpha = d->alpha; e.scale = d->scale; e.bias = d->bias; e.residual = nullptr; e.ldr = 0; e.mask = nullptr; e.ldmask = 0;
    e.act = d->act; e.atomic = 0; e.drop_scale = 0.0f; e.drop_thresh = 0; e.drop_seed = 0;
    e.vec = (!d->scale || aligned16(d->scale)) && (!d->bias || aligned16(d->bias));
    const bool bf = d->compute == 1;
    if (mode == 0) {
        DETR_REQUIRE(d->y_dtype == 0 || bf, "stem conv: a bf16 output needs compute = bf16");
        e.c16 = d->y_dtype == 1;
        a.w = d->w; a.y = d->y; a.e = e;
        dim3 grid((unsigned)cdiv(a.M, 64));
        if (bf) hipLaunchKernelGGL(stem_fwd_kernel<true>, grid, dim3(GEMM_THREADS), 0, s, a);
        else hipLaunchKernelGGL(stem_fwd_kernel<false>, grid, dim3(GEMM_THREADS), 0, s, a);
        DETR_LAUNCH_CHECK("stem conv forward");
        return 0;
    }
    // weight gradient: d->w = dy [M, 64], d->y = dw [147, 64] (accumulated: dw += alpha * scale[co] * sum)
    DETR_REQUIRE(!d->bias && d->act == 0, "stem conv wgrad: only scale/alpha epilogue");
    a.dy = d->w; a.dw = d->y;
    int split = d->split > 0 ? d->split : 1;
    const int bk = bf ? BF_BK : GEMM_BK;
    int rps = cdiv(cdiv(a.M, split), bk) * bk;
    split = cdiv(a.M, rps);
    a.rows_per_split = rps;
    const long long part = (long long)STEM_K * 64;
    const bool partial = split > 1 && d->workspace && aligned16(d->workspace) && d->workspace_bytes >= (long long)split * part * 4;
    DETR_REQUIRE(split == 1 || partial, "stem conv wgrad: split > 1 needs a workspace of split*147*64 floats");
    EpiArgs fin = e;
    if (partial) {
        a.dw = d->workspace;
        a.part_stride = part;
        e.alpha = 1.0f; e.scale = nullptr; e.vec = 1;
    } else {
        e.residual = d->y; e.ldr = 64;       // dw += ...
    }
    a.e = e;
    dim3 grid(3, 1, (unsigned)split);
    DETR_REQUIRE(d->w_dtype == 0 || bf, "stem conv wgrad: a bf16 dy needs compute = bf16");
    if (bf && d->w_dtype == 1) hipLaunchKernelGGL((stem_wgrad_kernel<true, true>), grid, dim3(GEMM_THREADS), 0, s, a);
    else if (bf) hipLaunchKernelGGL((stem_wgrad_kernel<true, false>), grid, dim3(GEMM_THREADS), 0, s, a);
    else hipLaunchKernelGGL((stem_wgrad_kernel<false, false>), grid, dim3(GEMM_THREADS), 0, s, a);
    DETR_LAUNCH_CHECK("stem conv wgrad");
    if (partial) {
        launch_splitk_reduce(d->workspace, split, part, STEM_K, 64, d->y, 64, fin.alpha, fin.scale, s);
        DETR_LAUNCH_CHECK("stem conv wgrad reduce");
    }
    return 0;
}
