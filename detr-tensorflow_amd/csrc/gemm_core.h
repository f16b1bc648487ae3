// gemm_core.h -- the fp32 MFMA tile engine shared by the plain GEMM and the implicit-GEMM
// 3x3 convolution kernels (gfx950 / CDNA4).
//
// Design (MI355X-first):
//   * v_mfma_f32_32x32x2_f32: exact f32 (bitwise an fmaf chain) at 157 TF peak; one f32 VGPR per
//     operand per lane, 16 accumulator registers per 32x32 tile.
//   * 256 threads = 4 waves arranged WGM x WGN; each wave owns TM x TN tiles of 32x32.
//   * operands are staged K-major in LDS ( S[k][m] ): a fragment read is one ds_read_b32 per
//     lane with lanes 0..31 on consecutive dwords of row k and lanes 32..63 on row k+1 ->
//     conflict free; rows are padded by 4 dwords so that the transposing ds_write_b32 of a
//     K-contiguous global operand is at most 2-way conflicted (free on ds_write_b32).
//   * global -> register -> LDS software pipeline, double-buffered LDS, ONE barrier per K tile:
//     the loads of tile t+1 are in flight while tile t is multiplied.
//   * workgroup -> tile mapping is XCD-aware (8 XCDs, private L2s): each XCD walks a contiguous
//     run of tiles with the N index fastest, so the A panel of a row block is fetched into one
//     L2 only.
#pragma once
#include "common.h"
#include <type_traits>

namespace detr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BK = 16;
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_PAD = 4;

struct EpiArgs {
    float alpha;
    const float *scale;
    const float *bias;
    const float *residual;
    long long ldr;
    const float *mask;
    long long ldmask;
    int act;     // 0 none 1 relu 2 sigmoid
    int atomic;  // 1: atomicAdd into C (split-K / wgrad)
    int vec;     // 1: C / residual / mask rows and scale / bias are 16-byte aligned -> float4 epilogue
    float drop_scale;        // 1/(1-p), 0 = no dropout; element index = row * N + col
    uint32_t drop_thresh;    // p * 2^16
    uint32_t drop_seed;      // dropout SITE id; the per-step seed is read from *drop_step (device memory, may be null = 0)
    // optional output-row remap (stride-2 conv dgrad, one launch per pixel-parity class): GEMM row r of the class
    // (n, h2, w2) addresses pixel (n, 2*h2 + remap_ph, 2*w2 + remap_pw) of the [.., remap_H, remap_W, N] tensors
    // C / residual / mask.  remap_w2 == 0: identity.
    int remap_w2 = 0, remap_h2 = 0, remap_W = 0, remap_H = 0, remap_ph = 0, remap_pw = 0;
    // bf16 STORAGE of the activation tensors (backbone, DETR_HIP_ACT16): C / residual / mask are bf16 in memory (uint16,
    // leading dimensions in elements); the arithmetic of the epilogue stays fp32, the result is rounded once (RNE)
    int c16 = 0, r16 = 0, m16 = 0;
    const uint32_t *drop_step = nullptr;
    // 1: all-bf16 epilogue streams (C, and residual / mask when present) with rows that are 16-byte aligned and a multiple of 8
    // wide, no atomics -> epilogue_wide16 (8 columns per lane, every request of a strip in flight at once)
    int wide16 = 0;
    // bit-packed ReLU masks (wide16 / streaming epilogues only): m16 == 2 -> `mask` points to bytes (bit (n & 7) of byte
    // [row * ldmask + n / 8], ldmask in BYTES); mbits_out != null -> the epilogue also writes those bytes for its own output
    unsigned char *mbits_out = nullptr;
    long long ld_mbits_out = 0;
};
// bit j of the result = (bf16 element j of the four packed pairs > 0)
__device__ __forceinline__ unsigned bf16x8_gt0_bits(unsigned w0, unsigned w1, unsigned w2, unsigned w3) {
    const unsigned w[4] = {w0, w1, w2, w3};
    unsigned b = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b |= ((short)(w[j] & 0xFFFFu) > 0 ? 1u : 0u) << (2 * j);
        b |= ((short)(w[j] >> 16) > 0 ? 1u : 0u) << (2 * j + 1);
    }
    return b;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ float4 ld_bf16x4(const void *base, long long elem) {
    const uint2 v = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(base) + elem);
    return make_float4(bf16_bits_to_f32(v.x & 0xFFFFu), bf16_bits_to_f32(v.x >> 16), bf16_bits_to_f32(v.y & 0xFFFFu),
                       bf16_bits_to_f32(v.y >> 16));
}
__device__ __forceinline__ float ld_bf16x1(const void *base, long long elem) {
    return bf16_bits_to_f32(reinterpret_cast<const unsigned short *>(base)[elem]);
}
// max(x, lo) as ONE v_max_f32: fmaxf() costs two under the IEEE mode of compute kernels (v_max x, x first, to quiet a signalling NaN
// that these epilogues never see -- their inputs are sums of finite products)
__device__ __forceinline__ float vmax_raw(float x, float lo) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(lo));
    return r;
}
__device__ __forceinline__ unsigned f32_to_bf16_pair(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}

template <int BM, int BN, int WGN>
struct SmemBytes;

template <int BM, int BN>
struct GemmSmem {
    static constexpr int LDA = BM + GEMM_PAD;
    static constexpr int LDB = BN + GEMM_PAD;
    float A[2][GEMM_BK][LDA];
    float B[2][GEMM_BK][LDB];
};

template <int BM, int BN, int WGN>
struct SmemBytes {
    static constexpr int TILES = (int)sizeof(GemmSmem<BM, BN>);
    static constexpr int STAGE = 4 * 32 * (BN / WGN + 4) * 4;
    static constexpr int VALUE = TILES > STAGE ? TILES : STAGE;
};

// bijective XCD-aware remap of a linear workgroup id (guide T1, bijective variant)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---------------------------------------------------------------------------------------------
// Operand access through a buffer descriptor (SRSRC): out-of-range lanes get the offset BUF_OOB, which
// the hardware bounds check turns into a zero result.  A guarded tile load is therefore one
// v_cndmask + buffer_load_dwordx4, with no exec-mask branch per load (the plain `if (ok) v = *p`
// form compiled to a saveexec/branch tree per float4 and a vmcnt(0) between the A and B loads).
// The descriptor base is wave-uniform (kernel argument + blockIdx-derived offsets); operands are
// < 4 GB (checked by the host dispatch) so byte offsets fit the 32-bit voffset.
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned BUF_OOB = 0xFFFFFFF0u;
constexpr long long BUF_MAX_BYTES = 0xFFFFFFF0ll;

struct BufSrc {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ void init(const float *base, long long elems) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)(unsigned)(elems * 4), 0x00020000);
    }
    __device__ __forceinline__ void init_bytes(const void *base, long long bytes) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(unsigned)bytes, 0x00020000);
    }
    __device__ __forceinline__ float4 ld4(unsigned voff) const {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
    }
    __device__ __forceinline__ uint4 ld16(unsigned voff) const {      // 16 raw bytes (8 bf16)
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
    }
    __device__ __forceinline__ uint4 ld16_nt(unsigned voff) const {   // 16 raw bytes of a read-once stream: non-temporal (aux = 2)
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 2));
    }
    __device__ __forceinline__ void st16(unsigned voff, u32x4 v) const {      // 16 raw bytes; an out-of-range offset (BUF_OOB) stores nothing
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 0);
    }
    __device__ __forceinline__ void st1(unsigned voff, unsigned char v) const {
        __builtin_amdgcn_raw_buffer_store_b8(v, rsrc, voff, 0, 0);
    }
    __device__ __forceinline__ uint2 ld8(unsigned voff) const {       // 8 raw bytes (4 bf16)
        return __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 0, 0));
    }
    __device__ __forceinline__ float ld1(unsigned voff) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0));
    }
    // four consecutive floats at byte offset voff, of which the first nvalid (may be <= 0 or > 4) exist
    __device__ __forceinline__ float4 ld4_vec(unsigned voff, int nvalid) const {
        return ld4(nvalid >= 4 ? voff : BUF_OOB);                // vec mode: extents are multiples of 4 (host check)
    }
    __device__ __forceinline__ float4 ld4_scalar(unsigned voff, int nvalid) const {
        float4 v;
        v.x = ld1(nvalid > 0 ? voff : BUF_OOB);
        v.y = ld1(nvalid > 1 ? voff + 4u : BUF_OOB);
        v.z = ld1(nvalid > 2 ? voff + 8u : BUF_OOB);
        v.w = ld1(nvalid > 3 ? voff + 12u : BUF_OOB);
        return v;
    }
};

// ---------------------------------------------------------------------------------------------
// Loader for an operand stored [mn][k] with k contiguous (row stride ld).
// Thread t handles float4 (row = (t>>2) + 64*i, k = (t&3)*4 .. +3) of the BMN x 16 tile.
// ---------------------------------------------------------------------------------------------
template <int BMN>
struct LoaderK {
    static constexpr int NV = (BMN >= 64) ? BMN / 64 : 1;
    BufSrc src;
    unsigned off[NV];      // byte offset of the row, BUF_OOB for rows outside the operand
    bool vec;
    int kq;
    int tid;

    // extent_elems > 0 overrides the descriptor size (conv: one descriptor over all 9 taps)
    __device__ __forceinline__ void init(const float *p, long long ld, int mn0, int MN, int K, bool vec_, int tid_,
                                         long long extent_elems = 0) {
        src.init(p, extent_elems > 0 ? extent_elems : (long long)(MN - 1) * ld + K);
        vec = vec_;
        tid = tid_;
        kq = (tid & 3) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = (tid >> 2) + 64 * i;
            const int g = mn0 + row;
            off[i] = ((row < BMN) && (g < MN)) ? (unsigned)((long long)g * ld * 4) : BUF_OOB;
        }
    }
    __device__ __forceinline__ void load(int k0, int K, float4 (&r)[NV], unsigned base = 0) const {
        const int k = k0 + kq;
        if (vec) {       // wave-uniform: one scalar branch per tile load, none per float4
#pragma unroll
            for (int i = 0; i < NV; ++i)
                r[i] = src.ld4_vec(off[i] + base + 4u * (unsigned)k, off[i] != BUF_OOB ? K - k : 0);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                r[i] = src.ld4_scalar(off[i] + base + 4u * (unsigned)k, off[i] != BUF_OOB ? K - k : 0);
        }
    }
    template <int LD>
    __device__ __forceinline__ void store(float (*S)[LD], const float4 (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = (tid >> 2) + 64 * i;
            if (row < BMN) {
                S[kq + 0][row] = r[i].x;
                S[kq + 1][row] = r[i].y;
                S[kq + 2][row] = r[i].z;
                S[kq + 3][row] = r[i].w;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Loader for an operand stored [k][mn] with mn contiguous (row stride ld).
// float4 index idx = t + 256*i : k row = idx / (BMN/4), column = (idx % (BMN/4))*4.
// ---------------------------------------------------------------------------------------------
template <int BMN>
struct LoaderMN {
    static constexpr int VPR = BMN / 4;
    static constexpr int TOTAL = GEMM_BK * VPR;
    static constexpr int NV = (TOTAL >= GEMM_THREADS) ? TOTAL / GEMM_THREADS : 1;
    BufSrc src;
    unsigned ld4b;         // row stride in bytes
    int mn0, MN;
    bool vec;
    int tid;

    __device__ __forceinline__ void init(const float *p, long long ld_, int mn0_, int MN_, int K, bool vec_, int tid_,
                                         long long extent_elems = 0) {
        src.init(p, extent_elems > 0 ? extent_elems : (long long)(K - 1) * ld_ + MN_);
        ld4b = (unsigned)(ld_ * 4);
        mn0 = mn0_;
        MN = MN_;
        vec = vec_;
        tid = tid_;
    }
    __device__ __forceinline__ void load(int k0, int K, float4 (&r)[NV], unsigned base = 0) const {
        unsigned o[NV];
        int nv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + GEMM_THREADS * i;
            const int kr = idx / VPR;
            const int c4 = (idx % VPR) * 4;
            const int k = k0 + kr;
            const int col = mn0 + c4;
            o[i] = base + (unsigned)k * ld4b + 4u * (unsigned)col;
            nv[i] = (idx < TOTAL && k < K) ? MN - col : 0;
        }
        if (vec) {
#pragma unroll
            for (int i = 0; i < NV; ++i) r[i] = src.ld4_vec(o[i], nv[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) r[i] = src.ld4_scalar(o[i], nv[i]);
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

// ---------------------------------------------------------------------------------------------
// One K tile of MFMAs out of LDS.
// MFMA 32x32x2 f32 operand map: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN>
struct TileCfg {
    static constexpr int WTM = BM / WGM;
    static constexpr int WTN = BN / WGN;
    static constexpr int TM = WTM / 32;
    static constexpr int TN = WTN / 32;
    static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 waves per workgroup (8: the ring kernel, gemm_ring.h)");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");
};

template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void mma_ktile(const float (*As)[BM + GEMM_PAD], const float (*Bs)[BN + GEMM_PAD],
                                          f32x16 (&acc)[TileCfg<BM, BN, WGM, WGN>::TM][TileCfg<BM, BN, WGM, WGN>::TN],
                                          int wm, int wn, int lane) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    const int l31 = lane & 31;
    const int kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < GEMM_BK; kk += 2) {
        float a[T::TM], b[T::TN];
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi) a[mi] = As[kk + kh][wm * T::WTM + mi * 32 + l31];
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni) b[ni] = Bs[kk + kh][wn * T::WTN + ni * 32 + l31];
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// The same K tile on the bf16 matrix pipe at fp32 accuracy ("f32x3", detr_gemm_desc.compute = 2).
// On gfx950 v_mfma_f32_32x32x2_f32 runs at 1/16 of the rate of v_mfma_f32_32x32x16_bf16.  An fp32 value is EXACTLY the sum of
// three bf16 values: h = bf16(x), m = bf16(x - h), l = x - h - m (both differences are exact in fp32, |m| <= 2^-8 |x|,
// |l| <= 2^-16 |x|, and l has at most 8 significant bits left).  A product x*y is then the sum of nine bf16 products, each
// exact in the fp32 accumulator's input; the three smallest (m*l', l*m', l*l': <= 2^-23 |xy| together, the size of one fp32
// rounding of the product) are dropped in the 6-term form (NT = 6), kept in the 9-term form (NT = 9).  The terms of one
// 32x32x16 block enter the fp32 accumulator smallest first: 6 (9) roundings of the accumulator per 16 k where the fp32 MFMA
// has 8.  6 bf16 MFMAs take 6 x 32 cycles against 8 x 64 for the same 16 k on the fp32 instruction: 2.67x (1.78x) the
// matrix-pipe rate, paid for with ~4.5 VALU instructions per fragment value (v_cvt_pk_bf16_f32, shift / mask, v_pk_add_f32).
// The 16-deep fp32 LDS tile of the exact kernels is ONE k-step of the bf16 instruction; a lane gathers its 8 consecutive k of
// row (lane & 31) with 8 ds_read_b32 (the same conflict-free pattern as mma_ktile: lanes 0..31 on consecutive dwords).
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
struct Split3Frag { bf16x8_t h, m, l; };
// RNE pair conversion as an opaque instruction: written as two scalar casts the compiler re-derives `pair << 16` from a second,
// single-value v_cvt_pk_bf16_f32 (7.5 instead of 4.5 VALU instructions per value in the ISA of the first version)
__device__ __forceinline__ unsigned cvt_pk_bf16_asm(f32x2_t x) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x[0]), "v"(x[1]));
    return r;
}
__device__ __forceinline__ f32x2_t bf16_pair_to_f32(unsigned p) {
    f32x2_t r;
    r[0] = __builtin_bit_cast(float, p << 16);
    r[1] = __builtin_bit_cast(float, p & 0xFFFF0000u);
    return r;
}
// (x[0], x[1]) -> the three bf16 pairs h, m, l with x = h + m + l exactly; 9 VALU instructions per pair (3 conversions, 4 unpack
// shifts / masks, 2 v_pk_add_f32)
#ifndef DETR_X3_ABLATE
#define DETR_X3_ABLATE 0         // timing experiments (results wrong): 1 = one MFMA term of six, 2 = no split arithmetic, 4 = no loop requests
#endif
__device__ __forceinline__ void split3_pair(f32x2_t x, unsigned &h, unsigned &m, unsigned &l) {
    if constexpr ((DETR_X3_ABLATE & 2) != 0) { h = cvt_pk_bf16_asm(x); m = h; l = h; return; }
    h = cvt_pk_bf16_asm(x);
    f32x2_t r = x - bf16_pair_to_f32(h);
    m = cvt_pk_bf16_asm(r);
    r = r - bf16_pair_to_f32(m);
    l = cvt_pk_bf16_asm(r);
}
__device__ __forceinline__ Split3Frag split3_frag(const f32x2_t (&f)[4]) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3_pair(f[j], h[j], m[j], l[j]);
    Split3Frag r;
    r.h = __builtin_bit_cast(bf16x8_t, make_uint4(h[0], h[1], h[2], h[3]));
    r.m = __builtin_bit_cast(bf16x8_t, make_uint4(m[0], m[1], m[2], m[3]));
    r.l = __builtin_bit_cast(bf16x8_t, make_uint4(l[0], l[1], l[2], l[3]));
    return r;
}
// the product terms of one 32x32x16 block, smallest first
template <int NT>
__device__ __forceinline__ f32x16 split3_mma(const Split3Frag &a, const Split3Frag &b, f32x16 c) {
    static_assert(NT == 6 || NT == 9, "6 or 9 product terms");
    if constexpr ((DETR_X3_ABLATE & 1) != 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    if constexpr (NT == 9) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.l, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.l, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.m, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    return c;
}
#ifndef DETR_SPLIT3_TERMS
#define DETR_SPLIT3_TERMS 6
#endif
template <int BM, int BN, int WGM, int WGN, int NT = DETR_SPLIT3_TERMS>
__device__ __forceinline__ void mma_ktile_split3(const float (*As)[BM + GEMM_PAD], const float (*Bs)[BN + GEMM_PAD],
                                                 f32x16 (&acc)[TileCfg<BM, BN, WGM, WGN>::TM][TileCfg<BM, BN, WGM, WGN>::TN],
                                                 int wm, int wn, int lane) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    static_assert(GEMM_BK == 16, "one fp32 K tile = one k-step of v_mfma_f32_32x32x16_bf16");
    const int l31 = lane & 31;
    const int k8 = (lane >> 5) * 8;
    // every fragment value is read first (one LDS round trip for the tile), then fragments are split in the order the MFMA blocks
    // need them: (a0, b0) -> block (0, 0) can issue while the vector pipe splits b1, a1, ...
    f32x2_t fa[T::TM][4], fb[T::TN][4];
#pragma unroll
    for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fa[mi][j][0] = As[k8 + 2 * j][wm * T::WTM + mi * 32 + l31];
            fa[mi][j][1] = As[k8 + 2 * j + 1][wm * T::WTM + mi * 32 + l31];
        }
#pragma unroll
    for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fb[ni][j][0] = Bs[k8 + 2 * j][wn * T::WTN + ni * 32 + l31];
            fb[ni][j][1] = Bs[k8 + 2 * j + 1][wn * T::WTN + ni * 32 + l31];
        }
    Split3Frag a[T::TM], b[T::TN];
    a[0] = split3_frag(fa[0]);
#pragma unroll
    for (int ni = 0; ni < T::TN; ++ni) {
        b[ni] = split3_frag(fb[ni]);
        acc[0][ni] = split3_mma<NT>(a[0], b[ni], acc[0][ni]);
    }
#pragma unroll
    for (int mi = 1; mi < T::TM; ++mi) {
        a[mi] = split3_frag(fa[mi]);
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni) acc[mi][ni] = split3_mma<NT>(a[mi], b[ni], acc[mi][ni]);
    }
}
// K tile of the fp32-storage kernels: exact fp32 MFMA, or the 3-way bf16 split (SPLIT3)
template <int BM, int BN, int WGM, int WGN, bool SPLIT3>
__device__ __forceinline__ void mma_ktile_sel(const float (*As)[BM + GEMM_PAD], const float (*Bs)[BN + GEMM_PAD],
                                              f32x16 (&acc)[TileCfg<BM, BN, WGM, WGN>::TM][TileCfg<BM, BN, WGM, WGN>::TN],
                                              int wm, int wn, int lane) {
    if constexpr (SPLIT3) mma_ktile_split3<BM, BN, WGM, WGN>(As, Bs, acc, wm, wn, lane);
    else mma_ktile<BM, BN, WGM, WGN>(As, Bs, acc, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------
// Epilogue. C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// The accumulators are transposed through LDS (one 32-row strip per wave at a time) so that every
// lane then owns 4 CONSECUTIVE columns of one row: residual / mask reads and the output store are
// 16-byte accesses and a wave instruction covers whole 256-byte row segments (the direct
// register->global form issued 4x more, 4-byte, store instructions and ran the output-heavy
// layer1/layer2 GEMMs at ~1 TB/s).
// ---------------------------------------------------------------------------------------------
template <int BN, int WGN>
struct StageCfg {
    static constexpr int WTN = BN / WGN;
    static constexpr int LD = WTN + 4;                 // floats; keeps rows 16-byte aligned
    static constexpr int FLOATS_PER_WAVE = 32 * LD;
    static constexpr int BYTES = 4 * FLOATS_PER_WAVE * 4;
};

// dropout position (transformer.py:169,174-176): with a residual, the GEMM result is dropped BEFORE the
// residual is added (x + drop(f(x))); without one, after the activation (drop(relu(.))).
__device__ __forceinline__ float epi_one(float v, float sc, float bi, const EpiArgs &e, float res, float msk, bool keep) {
    v = v * sc + bi;
    v *= e.alpha;
    const bool drop = e.drop_scale != 0.0f;
    if (drop && e.residual) v = keep ? v * e.drop_scale : 0.0f;
    v += res;
    if (e.act == 1) v = fmaxf(v, 0.0f);
    else if (e.act == 2) v = 1.0f / (1.0f + expf(-v));
    if (drop && !e.residual) v = keep ? v * e.drop_scale : 0.0f;
    if (e.mask) v = (msk > 0.0f) ? v : 0.0f;
    return v;
}

// All-bf16 form of the epilogue (EpiArgs.wide16).  Measured with the main loop compiled out (scripts/experiments, round 3): the
// rolled 4-column loop below is a chain of dependent round trips -- LDS read, residual / mask request, wait, store, next item --
// and at 8-12 waves per CU (128x128 tiles, the direct-A kernel) it ran the output streams at 1-2 TB/s: 30 us of the 67 us of
// M33600 N256 K1024 + mask, 100 us of 142 us for M33600 N1024 K512 + residual + mask.  Here a lane owns 8 consecutive columns
// (16-byte bf16 accesses, a wave instruction covers whole 128 / 256-byte row segments), the residual and mask rows of ALL passes
// of a 32-row strip are requested before the accumulators are even staged, and the per-column scale / bias are loaded once.
// The arithmetic per element is epi_one, unchanged: results are bit-identical to the 4-column form.
// residual / mask operands of the wide epilogues: read once by the launch.  DETR_EPI_NT = 1 requests them non-temporal (A/B builds; the streaming
// kernel gained 0.2 ms from the same hint, gemm_stream.h)
#ifndef DETR_EPI_NT
#define DETR_EPI_NT 0
#endif
__device__ __forceinline__ uint4 epi_ld16(const unsigned short *p) {
#if DETR_EPI_NT
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    const u32x4v v = __builtin_nontemporal_load(reinterpret_cast<const u32x4v *>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const uint4 *>(p);
#endif
}
template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void epilogue_wide16(const f32x16 (&acc)[TileCfg<BM, BN, WGM, WGN>::TM][TileCfg<BM, BN, WGM, WGN>::TN],
                                                float *stage_base, float *C, long long ldc, int M, int N, int m0, int n0,
                                                int wm, int wn, int lane, int wave, const EpiArgs &e) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    using S = StageCfg<BN, WGN>;
    constexpr int LPR = T::WTN / 8;                    // lanes per staged row
    constexpr int RPP = 64 / LPR;                      // rows per pass of the wave
    constexpr int NP = 32 / RPP;                       // passes per 32-row strip (WTN / 16)
    static_assert(LPR >= 2 && LPR <= 64 && NP >= 1, "wave tile width 16 .. 512");
    float *stage = stage_base + wave * S::FLOATS_PER_WAVE;
    const int l31 = lane & 31;
    const int rh = (lane >> 5) * 4;
    const int c8 = (lane % LPR) * 8;
    const int rsub = lane / LPR;
    const int col = n0 + wn * T::WTN + c8;
    const bool col_ok = col < N;                       // N % 8 == 0: the 8 columns are inside or outside together
    const uint32_t dkey = e.drop_scale != 0.0f ? drop_key(e.drop_seed, e.drop_step) : 0u;
    float sc[8], bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = 1.0f; bi[j] = 0.0f; }
    if (col_ok && e.scale) {
        const float4 a = *reinterpret_cast<const float4 *>(e.scale + col), b = *reinterpret_cast<const float4 *>(e.scale + col + 4);
        sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
    }
    if (col_ok && e.bias) {
        const float4 a = *reinterpret_cast<const float4 *>(e.bias + col), b = *reinterpret_cast<const float4 *>(e.bias + col + 4);
        bi[0] = a.x; bi[1] = a.y; bi[2] = a.z; bi[3] = a.w; bi[4] = b.x; bi[5] = b.y; bi[6] = b.z; bi[7] = b.w;
    }
    const unsigned short *res16 = reinterpret_cast<const unsigned short *>(e.residual);
    const unsigned short *msk16 = reinterpret_cast<const unsigned short *>(e.mask);
    unsigned short *C16 = reinterpret_cast<unsigned short *>(C);
    auto prow_of = [&](const int row) -> long long {      // output-row remap of the stride-2 input-gradient classes (EpiArgs)
        if (e.remap_w2 <= 0) return row;
        const int w2 = row % e.remap_w2, t2 = row / e.remap_w2;
        const int h2 = t2 % e.remap_h2, n2 = t2 / e.remap_h2;
        return ((long long)n2 * e.remap_H + 2 * h2 + e.remap_ph) * e.remap_W + 2 * w2 + e.remap_pw;
    };
#pragma unroll
    for (int mi = 0; mi < T::TM; ++mi) {
        const int rowbase = m0 + wm * T::WTM + mi * 32 + rsub;
        uint4 rr[NP], mm[NP];
        long long prow[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = rowbase + p * RPP;
            const bool ok = col_ok && row < M;
            prow[p] = prow_of(row);
            rr[p] = make_uint4(0u, 0u, 0u, 0u);
            mm[p] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
            if (ok && res16) rr[p] = epi_ld16(res16 + prow[p] * e.ldr + col);
            if (ok && msk16) {
                if (e.m16 == 2) mm[p].x = reinterpret_cast<const unsigned char *>(e.mask)[prow[p] * e.ldmask + (col >> 3)];      // 8 mask bits
                else mm[p] = epi_ld16(msk16 + prow[p] * e.ldmask + col);
            }
        }
        // the staging region is wave-private: only the first strip needs the workgroup (the main loop / a row-sum finish of the
        // other waves may still be reading the LDS it aliases); later strips and the write -> read turn-around are wave-local
        if (mi == 0) __syncthreads();
        else wave_lds_sync();
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[((r & 3) + 8 * (r >> 2) + rh) * S::LD + ni * 32 + l31] = acc[mi][ni][r];
        wave_lds_sync();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int rl = rsub + p * RPP;
            const int row = rowbase + p * RPP;
            if (!(col_ok && row < M)) continue;
            const float4 a0 = *reinterpret_cast<const float4 *>(stage + rl * S::LD + c8);
            const float4 a1 = *reinterpret_cast<const float4 *>(stage + rl * S::LD + c8 + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const unsigned rw[4] = {rr[p].x, rr[p].y, rr[p].z, rr[p].w};
            unsigned mw[4] = {mm[p].x, mm[p].y, mm[p].z, mm[p].w};
            if (msk16 && e.m16 == 2) {                 // expand the byte into the bf16 pairs the arithmetic below reads (1.0 / 0.0)
                const unsigned mb = mm[p].x;
#pragma unroll
                for (int j = 0; j < 4; ++j) mw[j] = ((mb >> (2 * j)) & 1u ? 0x3f80u : 0u) | ((mb >> (2 * j + 1)) & 1u ? 0x3f800000u : 0u);
            }
            bool keep[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) keep[j] = true;
            if (e.drop_scale != 0.0f) {                // element index row * N + col is a multiple of 8: four pair hashes
                const unsigned long long di = ((unsigned long long)prow[p] * N + col) >> 1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t h = drop_hash(dkey, di + j);
                    keep[2 * j] = (h & 0xFFFFu) >= e.drop_thresh;
                    keep[2 * j + 1] = (h >> 16) >= e.drop_thresh;
                }
            }
            unsigned ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float o0 = epi_one(av[2 * j], sc[2 * j], bi[2 * j], e, bf16_bits_to_f32(rw[j] & 0xFFFFu), bf16_bits_to_f32(mw[j] & 0xFFFFu), keep[2 * j]);
                const float o1 = epi_one(av[2 * j + 1], sc[2 * j + 1], bi[2 * j + 1], e, bf16_bits_to_f32(rw[j] >> 16), bf16_bits_to_f32(mw[j] >> 16), keep[2 * j + 1]);
                ow[j] = f32_to_bf16_pair(o0, o1);
            }
            *reinterpret_cast<uint4 *>(C16 + prow[p] * ldc + col) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            if (e.mbits_out) e.mbits_out[prow[p] * e.ld_mbits_out + (col >> 3)] = (unsigned char)bf16x8_gt0_bits(ow[0], ow[1], ow[2], ow[3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Tile-ordered split-K slabs (round 4).  A split's partial tile is private scratch that only the reduce launch ever reads, so
// it does not have to be row-major: every lane stores its accumulator quads as they sit in the MFMA C/D registers -- float4
// unit ((wave * TM + mi) * TN + ni) * 4 + q of the tile, lane-linear -- 16-byte stores that cover 1 KB per wave instruction, no
// LDS transposition, no barrier, no address arithmetic per element.  (Ablation, profiles/r02_ablation_tile_kernels.txt: with
// the K loop compiled out the row-major slab epilogue alone was 32 of the 58 us of M256 N1024 K33600.)  The reduce kernel
// (gemm_f32.hip: splitk_reduce_body) walks the slabs in the same unit order and un-permutes on its way to C: unit u of tile
// (tm, tn) holds rows tm*BM + wm*WTM + mi*32 + 8*q + 4*(lane >> 5) + {0,1,2,3} of column tn*BN + wn*WTN + ni*32 + (lane & 31).
// The slab of one split is tiles_m * tiles_n * BM * BN floats (edge tiles padded).
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void store_slab_ts(const f32x16 (&acc)[TileCfg<BM, BN, WGM, WGN>::TM][TileCfg<BM, BN, WGM, WGN>::TN],
                                              float *slab_tile, int wave, int lane) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    float4 *p = reinterpret_cast<float4 *>(slab_tile) + (size_t)wave * (T::TM * T::TN * 256) + lane;
#pragma unroll
    for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                p[((mi * T::TN + ni) * 4 + q) * 64] = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2],
                                                                  acc[mi][ni][4 * q + 3]);
}
// unit -> (row, column) inside a BM x BN tile of a 2 x 2 wave grid (the only grid the split-K kernels use)
__host__ __device__ __forceinline__ void slab_ts_unit(int u, int BM, int BN, int &row0, int &col) {
    const int TN = BN >> 6, TM = BM >> 6;
    const int lane = u & 63, q = (u >> 6) & 3;
    int rest = u >> 8;
    const int ni = rest % TN; rest /= TN;
    const int mi = rest % TM;
    const int w = rest / TM;
    row0 = (w >> 1) * (BM >> 1) + mi * 32 + 8 * q + 4 * (lane >> 5);
    col = (w & 1) * (BN >> 1) + ni * 32 + (lane & 31);
}

// ---------------------------------------------------------------------------------------------
// Row-complete epilogue with the LayerNorm fused (round 4; detr_gemm_desc.ln_fwd): the workgroup's tile spans ALL N = 256 columns
// (32 rows x 256, four waves side by side), so the rows of   C = drop((acc + bias) * alpha) + residual   are complete inside the
// workgroup and the LayerNormalization that follows every attention / FFN block of the transformer (transformer.py:151-152,
// 169-170,177, 215-233) runs on them while they are still on chip: the tile is staged row-major in LDS, one wave per row, a lane
// owns 4 consecutive columns -- the arithmetic, its order and the wave reductions are those of rowops.hip layernorm_fwd_kernel
// (C = 256: one float4 per lane), so y / mean / rstd equal the two-launch path bit for bit.  C (the LayerNorm INPUT, which the
// backward needs) is still written.  Saves the 800- / 8400-row LayerNorm launch behind the GEMM.
// ---------------------------------------------------------------------------------------------
struct LnArgs {
    const float *gamma = nullptr, *beta = nullptr;
    float *y = nullptr, *mean = nullptr, *rstd = nullptr;
    const float *add = nullptr;
    int add_rows = 0;
    float *y2 = nullptr;
    unsigned short *y16 = nullptr;
    float eps = 0.0f;
};
constexpr int LN_TILE_M = 32, LN_TILE_N = 256, LN_STAGE_LD = LN_TILE_N + 4;
__device__ __forceinline__ void epilogue_ln(const f32x16 (&acc)[1][2], float *stage, float *C, long long ldc, int M, int m0, int wn,
                                            int lane, int wave, const EpiArgs &e, const LnArgs &ln) {
    const int l31 = lane & 31, rh = (lane >> 5) * 4;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + rh) * LN_STAGE_LD + wn * 64 + ni * 32 + l31] = acc[0][ni][r];
    __syncthreads();
    const int c4 = lane * 4;
    const uint32_t dkey = e.drop_scale != 0.0f ? drop_key(e.drop_seed, e.drop_step) : 0u;
    float4 bi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.bias) bi = *reinterpret_cast<const float4 *>(e.bias + c4);
    const float4 gm = *reinterpret_cast<const float4 *>(ln.gamma + c4), bt = *reinterpret_cast<const float4 *>(ln.beta + c4);
#pragma unroll
    for (int rr = 0; rr < LN_TILE_M / 4; ++rr) {
        const int lr = wave + 4 * rr, row = m0 + lr;
        if (row >= M) continue;                                         // wave-uniform
        const float4 a = *reinterpret_cast<const float4 *>(stage + lr * LN_STAGE_LD + c4);
        float v[4] = {a.x, a.y, a.z, a.w};
        const float b4[4] = {bi.x, bi.y, bi.z, bi.w};
        float res[4] = {0.f, 0.f, 0.f, 0.f};
        if (e.residual) {
            const float4 r4 = *reinterpret_cast<const float4 *>(e.residual + (long long)row * e.ldr + c4);
            res[0] = r4.x; res[1] = r4.y; res[2] = r4.z; res[3] = r4.w;
        }
        bool keep[4] = {true, true, true, true};
        if (e.drop_scale != 0.0f) {                                     // element index row * N + col, one hash per pair (common.h)
            const unsigned long long idx = (unsigned long long)row * LN_TILE_N + (unsigned)c4;
            const uint32_t h0 = drop_hash(dkey, idx >> 1), h1 = drop_hash(dkey, (idx >> 1) + 1);
            keep[0] = (h0 & 0xFFFFu) >= e.drop_thresh; keep[1] = (h0 >> 16) >= e.drop_thresh;
            keep[2] = (h1 & 0xFFFFu) >= e.drop_thresh; keep[3] = (h1 >> 16) >= e.drop_thresh;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = epi_one(v[j], 1.0f, b4[j], e, res[j], 1.0f, keep[j]);
        *reinterpret_cast<float4 *>(C + (long long)row * ldc + c4) = make_float4(v[0], v[1], v[2], v[3]);
        // ---- layernorm_fwd_kernel, C = 256
        const float mu = wave_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)LN_TILE_N;
        const float d0 = v[0] - mu, d1 = v[1] - mu, d2 = v[2] - mu, d3 = v[3] - mu;
        const float var = wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) / (float)LN_TILE_N;
        const float rs = rsqrtf(var + ln.eps);
        float4 o;
        o.x = (v[0] - mu) * rs * gm.x + bt.x;
        o.y = (v[1] - mu) * rs * gm.y + bt.y;
        o.z = (v[2] - mu) * rs * gm.z + bt.z;
        o.w = (v[3] - mu) * rs * gm.w + bt.w;
        *reinterpret_cast<float4 *>(ln.y + (long long)row * LN_TILE_N + c4) = o;
        if (ln.y16) {
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            bf2 p0, p1;
            p0[0] = (__bf16)o.x; p0[1] = (__bf16)o.y; p1[0] = (__bf16)o.z; p1[1] = (__bf16)o.w;
            *reinterpret_cast<uint2 *>(ln.y16 + (long long)row * LN_TILE_N + c4) = make_uint2(__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1));
        }
        if (ln.y2) {
            const float4 p = *reinterpret_cast<const float4 *>(ln.add + (long long)(row % ln.add_rows) * LN_TILE_N + c4);
            *reinterpret_cast<float4 *>(ln.y2 + (long long)row * LN_TILE_N + c4) = make_float4(o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w);
        }
        if (lane == 0) {
            ln.mean[row] = mu;
            ln.rstd[row] = rs;
        }
    }
}

// ALLOW_WIDE = false: callers whose output can never take the all-bf16 form (weight gradients: fp32, accumulated) keep the
// 4-column code alone -- the second form would only cost them registers and code
template <int BM, int BN, int WGM, int WGN, bool ALLOW_WIDE = true>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[TileCfg<BM, BN, WGM, WGN>::TM][TileCfg<BM, BN, WGM, WGN>::TN],
                                         float *stage_base, float *C, long long ldc, int M, int N, int m0, int n0,
                                         int wm, int wn, int lane, int wave, const EpiArgs &e, const bool wg_sync = true) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    using S = StageCfg<BN, WGN>;
    if constexpr (ALLOW_WIDE) {
        if (e.wide16) {                                // (kernel argument: uniform over the grid)
            epilogue_wide16<BM, BN, WGM, WGN>(acc, stage_base, C, ldc, M, N, m0, n0, wm, wn, lane, wave, e);
            return;
        }
    }
    constexpr int VPR = T::WTN / 4;                    // float4 per staged row
    constexpr int ITERS = (32 * VPR) / 64;
    float *stage = stage_base + wave * S::FLOATS_PER_WAVE;
    const int l31 = lane & 31;
    const int rh = (lane >> 5) * 4;
    const int colbase = n0 + wn * T::WTN;
    const uint32_t dkey = e.drop_scale != 0.0f ? drop_key(e.drop_seed, e.drop_step) : 0u;
#pragma unroll
    for (int mi = 0; mi < T::TM; ++mi) {
        // (wave-private staging region: see epilogue_wide16; `wg_sync` = false: the caller's previous epilogue call on the same
        //  region already went through the workgroup barrier -- the nine taps of the fused weight gradient)
        if (mi == 0 && wg_sync) __syncthreads();
        else wave_lds_sync();
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[((r & 3) + 8 * (r >> 2) + rh) * S::LD + ni * 32 + l31] = acc[mi][ni][r];
        wave_lds_sync();
        const int rowbase = m0 + wm * T::WTM + mi * 32;
        // kept rolled on purpose: the body only touches LDS / global memory, and a small body lets the
        // compiler fully unroll the register-indexing mi / ni / r loops above (otherwise acc spills to scratch)
#pragma unroll 1
        for (int it = 0; it < ITERS; ++it) {
            const int idx = it * 64 + lane;
            const int rl = idx / VPR;
            const int c4 = (idx - rl * VPR) * 4;
            const int row = rowbase + rl;
            const int col = colbase + c4;
            if (row >= M || col >= N) continue;
            const float4 a = *reinterpret_cast<const float4 *>(stage + rl * S::LD + c4);
            long long prow = row;
            if (e.remap_w2 > 0) {
                const int w2 = row % e.remap_w2, t2 = row / e.remap_w2;
                const int h2 = t2 % e.remap_h2, n2 = t2 / e.remap_h2;
                prow = ((long long)n2 * e.remap_H + 2 * h2 + e.remap_ph) * e.remap_W + 2 * w2 + e.remap_pw;
            }
            float *dst = C + prow * ldc + col;                      // (fp32 output; the bf16 form is addressed below)
            unsigned short *dst16 = reinterpret_cast<unsigned short *>(C) + prow * ldc + col;
            if (e.vec && col + 3 < N) {
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), bi = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 rs = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
                if (e.scale) sc = *reinterpret_cast<const float4 *>(e.scale + col);
                if (e.bias) bi = *reinterpret_cast<const float4 *>(e.bias + col);
                if (e.residual) rs = e.r16 ? ld_bf16x4(e.residual, prow * e.ldr + col) : *reinterpret_cast<const float4 *>(e.residual + prow * e.ldr + col);
                if (e.mask) mk = e.m16 ? ld_bf16x4(e.mask, prow * e.ldmask + col) : *reinterpret_cast<const float4 *>(e.mask + prow * e.ldmask + col);
                float4 o;
                const unsigned long long di = (unsigned long long)prow * N + col;
                bool k0 = true, k1 = true, k2 = true, k3 = true;
                if (e.drop_scale != 0.0f) {
                    if ((di & 1ull) == 0) {      // two hashes serve the four elements (common.h: 16 bits per element)
                        const uint32_t h0 = drop_hash(dkey, di >> 1), h1 = drop_hash(dkey, (di >> 1) + 1);
                        k0 = (h0 & 0xFFFFu) >= e.drop_thresh; k1 = (h0 >> 16) >= e.drop_thresh;
                        k2 = (h1 & 0xFFFFu) >= e.drop_thresh; k3 = (h1 >> 16) >= e.drop_thresh;
                    } else {
                        k0 = drop_keep(dkey, di, e.drop_thresh); k1 = drop_keep(dkey, di + 1, e.drop_thresh);
                        k2 = drop_keep(dkey, di + 2, e.drop_thresh); k3 = drop_keep(dkey, di + 3, e.drop_thresh);
                    }
                }
                o.x = epi_one(a.x, sc.x, bi.x, e, rs.x, mk.x, k0);
                o.y = epi_one(a.y, sc.y, bi.y, e, rs.y, mk.y, k1);
                o.z = epi_one(a.z, sc.z, bi.z, e, rs.z, mk.z, k2);
                o.w = epi_one(a.w, sc.w, bi.w, e, rs.w, mk.w, k3);
                if (e.atomic) {
                    unsafeAtomicAdd(dst + 0, o.x);
                    unsafeAtomicAdd(dst + 1, o.y);
                    unsafeAtomicAdd(dst + 2, o.z);
                    unsafeAtomicAdd(dst + 3, o.w);
                } else if (e.c16) {
                    *reinterpret_cast<uint2 *>(dst16) = make_uint2(f32_to_bf16_pair(o.x, o.y), f32_to_bf16_pair(o.z, o.w));
                } else {
                    *reinterpret_cast<float4 *>(dst) = o;
                }
            } else {
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (col + j < N) {
                        const float sc = e.scale ? e.scale[col + j] : 1.0f;
                        const float bi = e.bias ? e.bias[col + j] : 0.0f;
                        const float rs = !e.residual ? 0.0f : (e.r16 ? ld_bf16x1(e.residual, prow * e.ldr + col + j) : e.residual[prow * e.ldr + col + j]);
                        const float mk = !e.mask ? 1.0f : (e.m16 ? ld_bf16x1(e.mask, prow * e.ldmask + col + j) : e.mask[prow * e.ldmask + col + j]);
                        const bool kp = e.drop_scale == 0.0f || drop_keep(dkey, (unsigned long long)prow * N + col + j, e.drop_thresh);
                        const float o = epi_one(av[j], sc, bi, e, rs, mk, kp);
                        if (e.atomic) unsafeAtomicAdd(dst + j, o);
                        else if (e.c16) dst16[j] = (unsigned short)(f32_to_bf16_pair(o, 0.0f) & 0xFFFFu);
                        else dst[j] = o;
                    }
                }
            }
        }
    }
}

}  // namespace detr
