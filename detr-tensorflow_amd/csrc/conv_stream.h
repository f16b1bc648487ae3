// conv_stream.h -- streaming 3x3 convolution for the 64-channel bottleneck convs of layer1 (resnet_backbone.py:122-126:
// stride 1, pad 1, Ci = Co = 64) in bf16 storage, forward and input gradient.
//
// Same construction as the streaming 1x1 GEMM (gemm_stream.h): the whole BN-folded kernel (9 taps x 64 x 64 bf16 = 73 KB)
// is staged in LDS once per workgroup as B[n][tap * 64 + k] (k contiguous), eight independent waves then walk over strips of
// 32 output pixels with NO workgroup barrier: the MFMA runs with swapped operands (D^T = W^T X^T), so a lane's A fragment
// is 16 contiguous bytes -- 8 channels of one input pixel of one tap -- loaded straight from global memory (the halo takes
// the buffer descriptor's out-of-range offset), requested one whole strip ahead; the accumulators go through a wave-private LDS
// transposition and leave as 16-byte stores of whole 128-byte pixels.  Against the implicit-GEMM tile kernel
// (conv_f32.hip) this removes the A tile's LDS round trip, both barriers per K tile and the per-tile loader arithmetic,
// which is what bounds that kernel (~75 instructions around 2 MFMAs per K tile).
//   forward : y[p][co] = relu(sum_{kh,kw,ci} x[p - 1 + (kh,kw)][ci] w[kh][kw][ci][co] + bias[co])
//   dgrad   : dx[p][ci] = (mask[p][ci] > 0) * sum_{kh,kw,co} dy[p + 1 - (kh,kw)][co] w[kh][kw][ci][co]
// Opt-in (DETR_HIP_CONV_STREAM=1): parity-tested (tests/test_gpu_kernels.py::test_conv3x3_stream64_opt_in) but measured at the
// same speed as the tile kernel in the full step, both with three taps in flight (22.80 vs 22.84 ms) and with the rolling
// one-strip-ahead prefetch below (22.77 vs 22.77 ms): two very different schedules landing on the same
// time says the bound is shared -- the 9x redundant, 16-bytes-per-lane input reads through the texture path are the
// suspect; the next step is counters (TA busy, L1 hit rate) and LDS-staged input rows.  Kept as the starting point.
#pragma once
#include "gemm_stream.h"

namespace detr {

struct ConvStreamArgs {
    int N, H, W, M;                                // M = N * H * W output pixels
    const unsigned short *src;                     // forward: x, dgrad: dy   [N, H, W, 64] bf16
    const unsigned short *w;                       // [3][3][64 ci][64 co] bf16
    unsigned short *dst;                           // forward: y, dgrad: dx   [N, H, W, 64] bf16
    const unsigned short *mask;                    // optional [N, H, W, 64] bf16: keeps dst where mask > 0
    const float *bias;                             // optional [64] fp32
    int act;                                       // 0 none, 1 ReLU
    int row_tiles;                                 // cdiv(M, 32)
};

constexpr int CS_WAVES = 8;
constexpr int CS_LDB = 9 * 64 + 8;                 // bf16 per staged kernel row (+8: fragment reads spread over the banks)
constexpr int CS_SMEM = 64 * CS_LDB * 2 + CS_WAVES * 32 * STREAM_LD * 4;

template <bool DGRAD, bool MASK>
__global__ __launch_bounds__(64 * CS_WAVES) void conv3x3_stream64_kernel(ConvStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char cs_smem[];
    unsigned short (*Bs)[CS_LDB] = reinterpret_cast<unsigned short (*)[CS_LDB]>(cs_smem);
    float *stage_all = reinterpret_cast<float *>(cs_smem + 64 * CS_LDB * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- prologue: kernel -> LDS as B[n][tap * 64 + k] -------------------------------------------------------------
    for (int c = tid; c < 9 * 64 * 8; c += 64 * CS_WAVES) {
        const int tap = c / 512, rem = c - tap * 512;
        const int ci = rem >> 3, co8 = (rem & 7) * 8;            // 16 bytes = 8 output channels of (tap, ci)
        const uint4 v = *reinterpret_cast<const uint4 *>(a.w + ((long long)tap * 64 + ci) * 64 + co8);
        if (DGRAD) {                                             // n = ci, k = co: contiguous as stored
            *reinterpret_cast<uint4 *>(&Bs[ci][tap * 64 + co8]) = v;
        } else {                                                 // n = co, k = ci: transposed
            const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Bs[co8 + 2 * i][tap * 64 + ci] = (unsigned short)(wv[i] & 0xFFFFu);
                Bs[co8 + 2 * i + 1][tap * 64 + ci] = (unsigned short)(wv[i] >> 16);
            }
        }
    }
    __syncthreads();

    const int l31 = lane & 31, hh = lane >> 5;
    const int erow = lane >> 3, ecg = lane & 7;
    float bias[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias[i] = a.bias ? a.bias[ecg * 8 + i] : 0.0f;
    BufSrc srcA, srcM;
    srcA.init_bytes(a.src, (long long)a.M * 64 * 2);
    if (MASK) srcM.init_bytes(a.mask, (long long)a.M * 64 * 2);
    float *stage = stage_all + wave * 32 * STREAM_LD;
    const int HW = a.H * a.W;

    // Rolling prefetch: the 36 fragment requests (9 taps x 4 k-steps) of a strip are issued one strip ahead, each into the
    // registers the MFMAs of the same tap of the current strip have just consumed -- every request has a whole strip
    // (72 MFMAs) to land, which is what the scattered, mostly HBM-latency input reads need at 2 waves per SIMD.
    struct Pix { int n, h, w; bool ok; };
    auto decode = [&](int rt) -> Pix {
        const int m = rt * 32 + l31;
        Pix p;
        p.ok = rt < a.row_tiles && m < a.M;
        const int mm = p.ok ? m : 0;
        p.n = mm / HW;
        const int rem = mm - p.n * HW;
        p.h = rem / a.W;
        p.w = rem - p.h * a.W;
        return p;
    };
    auto load_tap = [&](const Pix &p, int tap, uint4 (&f)[4]) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int hs = DGRAD ? p.h + 1 - kh : p.h - 1 + kh;
        const int ws = DGRAD ? p.w + 1 - kw : p.w - 1 + kw;
        const bool v = p.ok && hs >= 0 && hs < a.H && ws >= 0 && ws < a.W;
        const unsigned base = v ? (unsigned)(((p.n * a.H + hs) * a.W + ws) * 128 + hh * 16) : BUF_OOB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) f[kk] = srcA.ld16(base == BUF_OOB ? BUF_OOB : base + kk * 32);
    };
    const int stride = gridDim.x * CS_WAVES;
    int rt = blockIdx.x * CS_WAVES + wave;
    uint4 fa[9][4];
    {
        const Pix p0 = decode(rt);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) load_tap(p0, tap, fa[tap]);
    }
    for (; rt < a.row_tiles; rt += stride) {
        const int r0 = rt * 32;
        const Pix pn = decode(rt + stride);
        uint4 rmsk[4];
        if (MASK) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = r0 + it * 8 + erow;
                rmsk[it] = srcM.ld16(row < a.M ? (unsigned)(row * 128 + ecg * 16) : BUF_OOB);
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nh][r] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, fa[tap][kk]);
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const bf16x8 bfr = *reinterpret_cast<const bf16x8 *>(&Bs[nh * 32 + l31][tap * 64 + kk * 16 + hh * 8]);
                    acc[nh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, af, acc[nh], 0, 0, 0);
                }
            }
            load_tap(pn, tap, fa[tap]);                          // refill with the same tap of the next strip
        }
        // ---- wave-private transposition and epilogue (as in gemm_stream.h) ------------------------------------------
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(stage + l31 * STREAM_LD + nh * 32 + 8 * g + 4 * hh) =
                    make_float4(acc[nh][4 * g], acc[nh][4 * g + 1], acc[nh][4 * g + 2], acc[nh][4 * g + 3]);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + erow;
            const int row = r0 + rl;
            const float4 v0 = *reinterpret_cast<const float4 *>(stage + rl * STREAM_LD + ecg * 8);
            const float4 v1 = *reinterpret_cast<const float4 *>(stage + rl * STREAM_LD + ecg * 8 + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += bias[i];
            if (a.act == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.0f);
            }
            if (MASK) {
                float mk[8];
                stream_unpack8(rmsk[it], mk);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (mk[i] > 0.0f) ? v[i] : 0.0f;
            }
            if (row < a.M)
                *reinterpret_cast<uint4 *>(a.dst + (long long)row * 64 + ecg * 8) =
                    make_uint4(f32_to_bf16_pair(v[0], v[1]), f32_to_bf16_pair(v[2], v[3]), f32_to_bf16_pair(v[4], v[5]),
                               f32_to_bf16_pair(v[6], v[7]));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool DGRAD, bool MASK>
static int launch_conv_stream_cfg(const ConvStreamArgs &a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_stream64_kernel<DGRAD, MASK>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, CS_SMEM);
        if (e != hipSuccess) return -1;
        attr_set = true;
    }
    int grid = 256;                                              // one 8-wave workgroup per CU (144 KB of LDS)
    const int need = (a.row_tiles + CS_WAVES - 1) / CS_WAVES;
    if (grid > need) grid = need;
    hipLaunchKernelGGL((conv3x3_stream64_kernel<DGRAD, MASK>), dim3((unsigned)grid), dim3(64 * CS_WAVES), CS_SMEM, s, a);
    return 0;
}

static int launch_conv_stream(ConvStreamArgs a, bool dgrad, hipStream_t s) {
    a.row_tiles = (a.M + 31) / 32;
    const bool m = a.mask != nullptr;
    if (dgrad) return m ? launch_conv_stream_cfg<true, true>(a, s) : launch_conv_stream_cfg<true, false>(a, s);
    return m ? launch_conv_stream_cfg<false, true>(a, s) : launch_conv_stream_cfg<false, false>(a, s);
}

}  // namespace detr
