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
// forward on bf16 activations, input rows staged once (the halo idea of conv_halo.h).  The gathering kernel below issues
// 40 scalar dword loads (+ their k / 21 index arithmetic) per thread and 64 x 64 tile and runs at 1.15 TB/s / 123 TFLOP/s
// (rocprofv3: 326 us at B = 8, 800 x 1333) -- neither roofline.  Here a persistent workgroup keeps the kernel tensor in
// LDS for its lifetime and walks over tiles of 64 consecutive output pixels of one output row: the 7 input rows x 133
// input pixels x 3 channels the tile touches are loaded with coalesced dword loads (14 per thread), rounded to bf16 and
// staged as P[row][399]; output pixel m then finds the (kw, c) run of kernel row kh at P[kh][6 m ..]: with every kernel
// row padded from 21 to 24 entries (zero weights) an MFMA k-step of 8 never straddles rows and its fragment is 16 bytes
// at a 4-byte aligned LDS address (12 m + 2 j bytes, j in {0, 8, 16}; lane stride 3 dwords: conflict free).
// K' = 7 x 24 = 168 -> 176 (11 k-steps).
// ------------------------------------------------------------------------------------------------
constexpr int SR_TW = 64;                          // output pixels per tile
constexpr int SR_PE = (2 * SR_TW + 5) * 3;         // 399 input floats per staged row
constexpr int SR_PLD = 408;                        // bf16 per staged row: >= 6 * 63 + 24, the tail stays zero
constexpr int SR_KP = 176;                         // padded reduction depth
constexpr int SR_SLD = 36;                         // floats per staged output row (32 + 4)

struct StemRowsSmem {
    unsigned short P[2][7][SR_PLD];
    unsigned short B[SR_KP * 64];                  // transpose-read image of the padded kernel tensor (LoaderMNt<64> unit order)
    float stage[4][32][SR_SLD];
};

__global__ __launch_bounds__(GEMM_THREADS, 3) void stem_fwd_rows_bf16_kernel(StemArgs a, int tiles_w, int ntiles) {
    __shared__ __attribute__((aligned(16))) StemRowsSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    BufSrc src;
    src.init(a.img, (long long)a.N * a.H * a.W * 3);
    // ---- once per workgroup: zero tails of the patch rows, the padded kernel tensor as a transpose-read image
    for (int i = tid; i < 2 * 7 * (SR_PLD - SR_PE); i += GEMM_THREADS) {
        const int r = i / (SR_PLD - SR_PE), e = i - r * (SR_PLD - SR_PE);
        (&sm.P[0][0][0])[r * SR_PLD + SR_PE + e] = 0;
    }
    for (int u = tid; u < SR_KP * 16; u += GEMM_THREADS) {
        const int kp = u >> 4, col = (u & 15) * 4;
        const int kh = kp / 24, j = kp - 24 * kh;
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kh < 7 && j < 21) w = *reinterpret_cast<const float4 *>(a.w + (long long)(kh * 21 + j) * 64 + col);
        const int o = (((kp >> 2) * 4 + (col >> 4)) * 64) + (kp & 3) * 16 + ((col >> 2) & 3) * 4;
        *reinterpret_cast<uint2 *>(&sm.B[o]) = make_uint2(pack_bf16(w.x, w.y), pack_bf16(w.z, w.w));
    }
    // epilogue constants: a lane owns 8 consecutive channels of a pixel
    const int c8 = (lane & 3) * 8, rsub = lane >> 2;
    const int col = wn * 32 + c8;
    float bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bi[j] = a.e.bias ? a.e.bias[col + j] : 0.0f;
    // A fragment offsets (shorts, inside one patch buffer) of the 11 k-steps: k' = 16 s + 8 hi -> (kh, j)
    int aoff[11];
#pragma unroll
    for (int s = 0; s < 11; ++s) {
        const int kp = 16 * s + 8 * hi;
        const int kh = kp / 24, j = kp - 24 * kh;
        aoff[s] = (kh < 7 ? kh : 6) * SR_PLD + 6 * (wm * 32 + l31) + (kh < 7 ? j : 0);      // (k' >= 168: zero weights, any finite data)
    }
    float rp[14];
    auto patch_load = [&](int t, float (&rp)[14]) {                  // tile t >= ntiles: out-of-range offsets, no traffic
        const int twi = t % tiles_w;
        const int r2 = t / tiles_w;
        const int ho = r2 % a.Ho, n = r2 / a.Ho;
        const int fc0 = (2 * twi * SR_TW - 3) * 3;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int hin = 2 * ho - 3 + r;
            const bool rok = t < ntiles && hin >= 0 && hin < a.H;
            const unsigned rowb = (unsigned)((n * a.H + hin) * a.W * 3 + fc0) * 4u;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int e = tid + 256 * h2;
                const bool ok = rok && e < SR_PE && fc0 + e >= 0 && fc0 + e < 3 * a.W;
                rp[2 * r + h2] = src.ld1(ok ? rowb + 4u * (unsigned)e : BUF_OOB);
            }
        }
    };
    const int e1c = tid + 256 < SR_PLD - 1 ? tid + 256 : SR_PLD - 1;
    auto patch_store = [&](int buf, const float (&rp)[14]) {
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            // (unconditional: elements past the 399 of a row were requested out of range = 0 and land in the zero tail)
            sm.P[buf][r][tid] = (unsigned short)(pack_bf16(rp[2 * r], 0.0f) & 0xFFFFu);
            sm.P[buf][r][e1c] = (unsigned short)(pack_bf16(rp[2 * r + 1], 0.0f) & 0xFFFFu);
        }
    };
    int t = blockIdx.x;
    patch_load(t, rp);
    patch_store(0, rp);
    patch_load(t + gridDim.x, rp);
    __syncthreads();
    int cur = 0;
    for (; t < ntiles; t += gridDim.x) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const unsigned short *P = &sm.P[cur][0][0];
#pragma unroll
        for (int s = 0; s < 11; ++s) {
            const uint32_t *pa = reinterpret_cast<const uint32_t *>(P + aoff[s]);
            const bf16x8 fa = __builtin_bit_cast(bf16x8, make_uint4(pa[0], pa[1], pa[2], pa[3]));
            const bf16x8 fb = frag_tr<64>(reinterpret_cast<const unsigned short (*)[BF_LD]>(sm.B + (s >> 1) * 2048), wn * 32, (s & 1) * 16, lane);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
        // the next tile's rows (requested one tile ago) go to the other buffer; request the tile after it
        patch_store(cur ^ 1, rp);
        patch_load(t + 2 * gridDim.x, rp);
        // ---- epilogue: wave-private transposition, y = relu(acc + shift) as bf16, 16 bytes per lane
        const int twi = t % tiles_w;
        const int r2 = t / tiles_w;
        const long long prow0 = (long long)r2 * a.Wo + twi * SR_TW + wm * 32;      // r2 = n * Ho + ho
        const int wn_ok = a.Wo - (twi * SR_TW + wm * 32);
        float *stage = &sm.stage[wave][0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * hi) * SR_SLD + l31] = acc[r];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int tw = it * 16 + rsub;
            const float4 v0 = *reinterpret_cast<const float4 *>(stage + tw * SR_SLD + c8);
            const float4 v1 = *reinterpret_cast<const float4 *>(stage + tw * SR_SLD + c8 + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] += bi[j];
                if (a.e.act == 1) v[j] = fmaxf(v[j], 0.0f);
            }
            if (tw < wn_ok)
                *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(a.y) + (prow0 + tw) * 64 + col) =
                    make_uint4(f32_to_bf16_pair(v[0], v[1]), f32_to_bf16_pair(v[2], v[3]), f32_to_bf16_pair(v[4], v[5]), f32_to_bf16_pair(v[6], v[7]));
        }
        __syncthreads();                           // patch[cur ^ 1] complete, patch[cur] and the stage free
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// forward: y = act((gather(img) @ w) + bias)      tile 64 pixels x 64 channels
// ------------------------------------------------------------------------------------------------
// MODE: 0 = exact fp32 MFMA, 1 = bf16 MFMA, 2 = fp32 accuracy on the bf16 matrix pipe (gemm_core.h: mma_ktile_split3)
template <int MODE>
__global__ __launch_bounds__(GEMM_THREADS) void stem_fwd_kernel(StemArgs a) {
    constexpr bool BF = MODE == 1;
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
            mma_ktile_sel<BM, BN, 2, 2, MODE == 2>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
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
template <int MODE, bool D16>
__global__ __launch_bounds__(GEMM_THREADS) void stem_wgrad_kernel(StemArgs a) {
    constexpr bool BF = MODE == 1;
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
            mma_ktile_sel<BM, BN, 2, 2, MODE == 2>(sm.A[cur], sm.B[cur], acc, wm, wn, lane);
            if (more) {
                store_a(sm.A[cur ^ 1]);
                lb.store(sm.B[cur ^ 1], rb);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    float *dw = a.dw + (long long)blockIdx.z * a.part_stride;
    epilogue<BM, BN, 2, 2, false>(acc, reinterpret_cast<float *>(smem_raw), dw, 64, STEM_K, 64, k0, 0, wm, wn, lane, wave, a.e);
}

// ------------------------------------------------------------------------------------------------
// weight gradient on bf16 activations with the input rows staged once (see stem_fwd_rows_bf16_kernel): a persistent
// workgroup accumulates ALL 176 padded kernel rows x 64 channels (12 MFMA tiles, 3 per wave) over units of 32 consecutive
// output pixels.  Per unit the 7 x 69-pixel input rows are staged as P[row][207] and dy[32][64] as a transpose-read image;
// the A^T fragment of kernel row k' = (kh, j) -- 8 consecutive output pixels m -- is 8 two-byte LDS reads at a 12-byte
// stride (P[kh][6 m + j]), no index arithmetic.  The gathering kernel above needs 3 workgroups (64 k each) x 8 scalar global
// gathers per thread and unit for the same operand (rocprofv3: 322 us at B = 8, 800 x 1333, 1.1 TB/s).
// Every workgroup writes its partial [147][64] slab; the caller reduces them (launch_splitk_reduce).
// ------------------------------------------------------------------------------------------------
constexpr int SW_TW = 32;
constexpr int SW_PE = (2 * SW_TW + 5) * 3;         // 207
constexpr int SW_PLD = 216;                        // >= 6 * 31 + 24, the tail stays zero

struct StemWgradRowsSmem {
    unsigned short P[2][7][SW_PLD];
    unsigned short D[2][32 * 64];
};

__global__ __launch_bounds__(GEMM_THREADS, 3) void stem_wgrad_rows_bf16_kernel(StemArgs a, int chunks, int nunits) {
    __shared__ __attribute__((aligned(16))) StemWgradRowsSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave & 1, rb0 = wave >> 1;      // this wave: channel block cb, kernel-row blocks rb0, rb0 + 2, rb0 + 4
    const int l31 = lane & 31, hi = lane >> 5;
    BufSrc src, dsrc;
    src.init(a.img, (long long)a.N * a.H * a.W * 3);
    dsrc.init_bytes(a.dy, (long long)a.M * 64 * 2);
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    int aoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int kp = (rb0 + 2 * i) * 32 + l31;
        const int kh = kp / 24, j = kp - 24 * kh;
        aoff[i] = (kh < 7 ? kh * SW_PLD + j : 6 * SW_PLD) + 48 * hi;      // (k' >= 168: rows nobody stores, any finite data)
    }
    const int e0c = tid < SW_PLD - 1 ? tid : SW_PLD - 1;
    float rp[7];
    uint2 rd[2];
    auto unit_load = [&](int u, float (&rp)[7], uint2 (&rd)[2]) {       // u >= nunits: out-of-range offsets, no traffic
        const int chunk = u % chunks;
        const int r2 = u / chunks;                 // n * Ho + ho
        const int ho = r2 % a.Ho, n = r2 / a.Ho;
        const int wo0 = chunk * SW_TW;
        const int fc0 = (2 * wo0 - 3) * 3;
        const bool live = u < nunits;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int hin = 2 * ho - 3 + r;
            const bool ok = live && hin >= 0 && hin < a.H && tid < SW_PE && fc0 + tid >= 0 && fc0 + tid < 3 * a.W;
            rp[r] = src.ld1(ok ? (unsigned)((n * a.H + hin) * a.W * 3 + fc0 + tid) * 4u : BUF_OOB);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + 256 * i;
            const int j = 4 * (v >> 6) + ((v >> 2) & 3);
            const int col = 16 * ((v >> 4) & 3) + 4 * (v & 3);
            const bool ok = live && wo0 + j < a.Wo;
            rd[i] = dsrc.ld8(ok ? ((unsigned)(r2 * a.Wo + wo0 + j) * 64u + (unsigned)col) * 2u : BUF_OOB);
        }
    };
    auto unit_store = [&](int buf, const float (&rp)[7], const uint2 (&rd)[2]) {
#pragma unroll
        for (int r = 0; r < 7; ++r) sm.P[buf][r][e0c] = (unsigned short)(pack_bf16(rp[r], 0.0f) & 0xFFFFu);   // tid >= 207: zeros into the tail
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<uint2 *>(&sm.D[buf][(tid + 256 * i) * 4]) = rd[i];
    };
    int u = blockIdx.x;
    unit_load(u, rp, rd);
    unit_store(0, rp, rd);
    unit_load(u + gridDim.x, rp, rd);
    __syncthreads();
    int cur = 0;
    for (; u < nunits; u += gridDim.x) {
        unit_store(cur ^ 1, rp, rd);
        unit_load(u + 2 * gridDim.x, rp, rd);
        const unsigned short *P = &sm.P[cur][0][0];
        const unsigned short(*D)[BF_LD] = reinterpret_cast<const unsigned short(*)[BF_LD]>(sm.D[cur]);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const bf16x8 fb = frag_tr<64>(D, cb * 32, st * 16, lane);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const unsigned short *pp = P + aoff[i] + st * 96;
                const uint4 w = make_uint4((uint32_t)pp[0] | ((uint32_t)pp[6] << 16), (uint32_t)pp[12] | ((uint32_t)pp[18] << 16),
                                           (uint32_t)pp[24] | ((uint32_t)pp[30] << 16), (uint32_t)pp[36] | ((uint32_t)pp[42] << 16));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), fb, acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    float *ws = a.dw + (long long)blockIdx.x * a.part_stride;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kp = (rb0 + 2 * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int kh = kp / 24, j = kp - 24 * kh;
            if (kh < 7 && j < 21) ws[(kh * 21 + j) * 64 + cb * 32 + l31] = acc[i][r];
        }
}

}  // namespace detr

using namespace detr;

extern "C" int64_t detr_hip_workspace_bytes_stem(const detr_stem_desc *d, int32_t mode) {
    if (!d || d->N <= 0 || d->Ho <= 0 || d->Wo <= 0) return -1;
    if (mode != 2) return 0;
    const long long part = (long long)STEM_K * 64;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    if (d->compute == 1 && d->w_dtype == 1 && M * 64 * 2 <= BUF_MAX_BYTES && tune(T_STEM_ROWS) != 2) {
        const int nunits = d->N * d->Ho * cdiv(d->Wo, SW_TW);        // the row-staging kernel: one partial per workgroup
        return (int64_t)(nunits < 768 ? nunits : 768) * part * 4;
    }
    int split = d->split > 0 ? d->split : 1;
    const int bk = d->compute == 1 ? BF_BK : GEMM_BK;
    const int rps = cdiv(cdiv(M, split), bk) * bk;
    split = cdiv(M, rps);
    return split > 1 ? (int64_t)split * part * 4 : 0;
}

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
        // bf16 output, folded scale: the row-staging kernel (DETR_HIP_STEM_ROWS=2: the gathering kernel)
        if (bf && e.c16 && !d->scale && d->alpha == 1.0f && (d->act == 0 || d->act == 1) && (!d->bias || aligned16(d->bias)) &&
            tune(T_STEM_ROWS) != 2) {
            const int tiles_w = cdiv(a.Wo, SR_TW), ntiles = a.N * a.Ho * tiles_w;
            const int wgs = ntiles < 768 ? ntiles : 768;
            hipLaunchKernelGGL(stem_fwd_rows_bf16_kernel, dim3((unsigned)wgs), dim3(GEMM_THREADS), 0, s, a, tiles_w, ntiles);
            DETR_LAUNCH_CHECK("stem conv forward (staged rows)");
            return 0;
        }
        if (bf) hipLaunchKernelGGL(stem_fwd_kernel<1>, grid, dim3(GEMM_THREADS), 0, s, a);
        else if (d->compute == 2) hipLaunchKernelGGL(stem_fwd_kernel<2>, grid, dim3(GEMM_THREADS), 0, s, a);
        else hipLaunchKernelGGL(stem_fwd_kernel<0>, grid, dim3(GEMM_THREADS), 0, s, a);
        DETR_LAUNCH_CHECK("stem conv forward");
        return 0;
    }
    // weight gradient: d->w = dy [M, 64], d->y = dw [147, 64] (accumulated: dw += alpha * scale[co] * sum)
    DETR_REQUIRE(!d->bias && d->act == 0, "stem conv wgrad: only scale/alpha epilogue");
    a.dy = d->w; a.dw = d->y;
    {   // bf16 dy: the row-staging kernel with its own persistent grid (DETR_HIP_STEM_ROWS=2: the gathering kernel)
        const int chunks = cdiv(a.Wo, SW_TW), nunits = a.N * a.Ho * chunks;
        const int wgs = nunits < 768 ? nunits : 768;
        const long long part = (long long)STEM_K * 64;
        if (bf && d->w_dtype == 1 && d->workspace && aligned16(d->workspace) && d->workspace_bytes >= (long long)wgs * part * 4 &&
            (long long)a.M * 64 * 2 <= BUF_MAX_BYTES && tune(T_STEM_ROWS) != 2) {
            a.dw = d->workspace;
            a.part_stride = part;
            hipLaunchKernelGGL(stem_wgrad_rows_bf16_kernel, dim3((unsigned)wgs), dim3(GEMM_THREADS), 0, s, a, chunks, nunits);
            DETR_LAUNCH_CHECK("stem conv wgrad (staged rows)");
            launch_splitk_reduce(d->workspace, wgs, part, STEM_K, 64, d->y, 64, e.alpha, e.scale, s);
            DETR_LAUNCH_CHECK("stem conv wgrad reduce");
            return 0;
        }
    }
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
    if (bf && d->w_dtype == 1) hipLaunchKernelGGL((stem_wgrad_kernel<1, true>), grid, dim3(GEMM_THREADS), 0, s, a);
    else if (bf) hipLaunchKernelGGL((stem_wgrad_kernel<1, false>), grid, dim3(GEMM_THREADS), 0, s, a);
    else if (d->compute == 2) hipLaunchKernelGGL((stem_wgrad_kernel<2, false>), grid, dim3(GEMM_THREADS), 0, s, a);
    else hipLaunchKernelGGL((stem_wgrad_kernel<0, false>), grid, dim3(GEMM_THREADS), 0, s, a);
    DETR_LAUNCH_CHECK("stem conv wgrad");
    if (partial) {
        launch_splitk_reduce(d->workspace, split, part, STEM_K, 64, d->y, 64, fin.alpha, fin.scale, s);
        DETR_LAUNCH_CHECK("stem conv wgrad reduce");
    }
    return 0;
}
