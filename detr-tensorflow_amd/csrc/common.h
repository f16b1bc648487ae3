// common.h -- shared host/device helpers of libdetr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/detr_hip.h"

namespace detr {

void set_error(const char *fmt, ...);

#define DETR_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::detr::set_error(__VA_ARGS__);     \
            return -1;                          \
        }                                       \
    } while (0)

#define DETR_LAUNCH_CHECK(name)                                                         \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            ::detr::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return -2;                                                                  \
        }                                                                               \
    } while (0)

// out[r*ldc + c] += alpha * scale[c] * sum_s ws[s*part_stride + r*cols + c]   (gemm_f32.hip)
// ts_bm > 0: tile-ordered slabs (gemm_core.h: store_slab_ts), tiles of ts_bm x ts_bn, ts_tiles_n per tile row
void launch_splitk_reduce(const float *ws, int splits, long long part_stride, int rows, int cols, float *C, long long ldc,
                          float alpha, const float *scale, hipStream_t stream, const float *rs_ws = nullptr,
                          float *rs_out = nullptr, float rs_alpha = 1.0f, int ts_bm = 0, int ts_bn = 0, int ts_tiles_n = 0);

// Minimum waves per SIMD requested for the 64x64-tile bf16 GEMM / conv kernels (second __launch_bounds__ argument):
// tuning knob (make DEFS=-DDETR_GEMM64_MINW=6).  Measured: 6 (<= 64 VGPRs, 8 workgroups / CU instead of 5) buys the
// HBM-bound short-K GEMMs 1-2 % and costs the long-K ones 10-25 % (tighter scheduling) -- left at 1.
#ifndef DETR_GEMM64_MINW
#define DETR_GEMM64_MINW 1
#endif

// Ablation builds for timing experiments only (scripts/experiments/ablate.sh; results are WRONG with any bit set): bit 0 = no
// MFMA issue, bit 1 = no global operand requests inside the K loop, bit 2 = no LDS operand stores inside the K loop, bit 3 = no
// workgroup barrier inside the K loop, bit 4 = no LDS fragment reads; gemm_stream.h: bit 5 = no residual / mask requests,
// bit 6 = no output stores.  The product library is built with 0.
#ifndef DETR_ABLATE
#define DETR_ABLATE 0
#endif
// 1: the bf16 tile GEMM K loop in its explicitly pipelined form (gemm_bf16_core.h: KPipe); 0: compiler-scheduled (round 3) -- A/B builds
#ifndef DETR_KLOOP_PIPE
#define DETR_KLOOP_PIPE 1
#endif

#if defined(__HIPCC__)
template <typename T>
__device__ __forceinline__ void ablate_keep(const T &v) {      // keeps a loaded register alive without using it
    const unsigned *p = reinterpret_cast<const unsigned *>(&v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) asm volatile("" ::"v"(p[i]));
}
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release / acquire of ALL address
// spaces, and on gfx9 loads and stores share vmcnt -- it therefore drains every outstanding global load, which would
// serialise an operand prefetch that is meant to stay in flight across the barrier.
#if defined(__HIPCC__)
// Ordering of LDS accesses INSIDE one wave (a wave-private staging region written by some lanes and read by others): the wave runs in
// lockstep, so a fence on the LDS counter is enough -- no workgroup barrier, the other waves are not involved.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#endif

// Tuning hooks: the integer environment variables DETR_HIP_<NAME> (0 when unset).  They are read ONCE, when the library is
// loaded -- never on the launch path (getenv is not safe against a concurrent setenv, and a launch should not cost a dozen
// environment scans) -- and again only on an explicit detr_hip_reload_tuning() (tests / tuning scripts that change a
// variable inside the process).
enum TuneKey {
    T_GEMM_TILE,
    T_SPLIT_XCD,
    T_GEMM_STREAM,
    T_GEMM_GROUP,
    T_STREAM_SL,
    T_STREAM_NW,
    T_CONV_HALO,
    T_CONV_TILE,
    T_DGRAD_S2_CLASSES,
    T_WGRAD_FUSED,
    T_WGRAD_FUSED_WGS,
    T_WGRAD_TILE,
    T_STEM_ROWS,
    T_ATTN_WAVES,
    T_ATTN_SPLIT,
    T_GEMM_K64,
    T_EPI_WIDE,
    T_SLAB_TS,           // 2: split-K slabs stay row-major (A/B of the tile-ordered form, round 4)
    T_GEMM_RING,         // 2: the 8-wave LDS-DMA ring kernel (gemm_ring.h, round 5) is off; 1: taken wherever eligible (tests)
    T_RING_NS,           // ring stages forced (2 / 3); 0 = as many as fit
    T_RING_BN,           // column panel forced (128 / 256); 0 = cost model
    T_RING_WGS,          // workgroup target forced; 0 = cost model
    T_RING_ROWS,         // row pitch forced (sweeps)
    T_RING_WTILE,        // ring weight gradient on 128 x 256 (1) / 256 x 128 (2) tiles (experiments)
    T_RING_ABLATE,       // timing experiments: RingArgs.ablate bits (results wrong when set)
    T_CONV_DMA,          // 2: the LDS-DMA form of the halo 3x3 kernel (conv_halo_dma.h, round 5) is off; 3: its requests in front of the fragment reads
    T_SPLIT3_T128,       // compute = 2 (f32x3): 128 x 128 tiles from this many of them on (0 = default rule)
    T_X3_DB,             // f32x3 GEMM (gemm_x3.h): 1 = double-buffered LDS form for every tile, 2 = single-buffered for every tile, 0 = rule
    T_X3_T192,           // f32x3 GEMM / 3x3 convolution: 2 = 64 x 64 and 128 x 128 tiles only, 1 = 192 x 128 wherever eligible, 3 = (convolution) 128 x 64 wherever
                         // 128 x 128 would run, 0 = the round-count rules (gemm_pick_tile, detr_hip_conv3x3_f32)
    T_X3_WG_ROUNDS,      // f32x3 3x3 weight gradient: rounds of workgroups the split plan aims at (0 = 2)
    T_X3_CONV,           // f32x3 3x3 convolutions: 2 = the per-wave split of the first form (conv3x3_kernel<.., SPLIT3>) everywhere, 1 = split once into LDS (conv_x3.h) everywhere, 0 = rule
    T_SPLIT3_ALL,        // compute = 2: 1 = every 64x64 / 128x128 GEMM launch takes the split kernel (tests), 0 = where it is faster
    T_COUNT
};
int tune(TuneKey k);

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Counter-based dropout: element `idx` of dropout site `site` in training step `*step` is kept iff 16 bits of a
// keyed 32-bit integer hash are >= thresh16 = p * 2^16.  Forward and backward kernels regenerate the same mask from
// (site, step seed, idx); tests/oracle restate the same hash in numpy (Keras Dropout semantics: kept values are
// scaled by 1/(1-p); the reference's TF RNG stream itself is not reproducible).
//   key  = mix32(step_seed + site * 0x9E3779B9)      -- step_seed is read from DEVICE memory (hipGraph replays of a
//          captured training step see a new seed without re-capturing); the host mixes (base seed, step, DP rank) into it
//   hash = two-round multiply-xorshift of the pair index with `key` injected before the first round and a second,
//          key-derived word injected between the rounds: two sites / steps / ranks are NOT related by an XOR
//          permutation of one random field (round-1 defect: mask_B(i) == mask_A(i ^ (A ^ B))).
// One 32-bit hash serves the element PAIR (idx & ~1, idx | 1): the low / high 16 bits are compared with
// thresh16 = p * 2^16 (p = 0.1 -> 6553/65536).  Kernels whose lanes own adjacent elements (attention
// probabilities, float4 epilogues) therefore evaluate one hash per two elements.
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t drop_key(uint32_t site, const uint32_t *step) {
    return mix32((step ? *step : 0u) + site * 0x9E3779B9u);
}
__host__ __device__ __forceinline__ uint32_t drop_hash(uint32_t key, unsigned long long pair) {
    const uint32_t key2 = key * 0x85EBCA6Bu + 0xC2B2AE35u;      // wave-uniform: scalar ALU
    uint32_t x = (uint32_t)pair ^ key;
    x ^= (uint32_t)(pair >> 32) * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= key2;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ bool drop_keep(uint32_t key, unsigned long long idx, uint32_t thresh16) {
    const uint32_t h = drop_hash(key, idx >> 1);
    return ((idx & 1ull) ? (h >> 16) : (h & 0xFFFFu)) >= thresh16;
}
static inline uint32_t drop_thresh16(float p) { return (uint32_t)(p * 65536.0f); }

// wave64 reductions (all 64 lanes participate; the result is wave-uniform).  DPP row-shift scan inside each row of
// 16 lanes, then row_bcast:15 / row_bcast:31 fold the four rows into lane 63 (the gfx9 wave64 reduction idiom):
// 6 DPP-modified VALU ops + one v_readlane instead of 6 dependent ds_bpermute round trips through the LDS pipe.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_f32(float identity, float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, x),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov_f32<0x111, 0xf>(0.0f, v);   // row_shr:1
    v += dpp_mov_f32<0x112, 0xf>(0.0f, v);   // row_shr:2
    v += dpp_mov_f32<0x114, 0xf>(0.0f, v);   // row_shr:4
    v += dpp_mov_f32<0x118, 0xf>(0.0f, v);   // row_shr:8
    v += dpp_mov_f32<0x142, 0xa>(0.0f, v);   // row_bcast:15 into rows 1 and 3
    v += dpp_mov_f32<0x143, 0xc>(0.0f, v);   // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    const float ninf = -INFINITY;
    v = fmaxf(v, dpp_mov_f32<0x111, 0xf>(ninf, v));
    v = fmaxf(v, dpp_mov_f32<0x112, 0xf>(ninf, v));
    v = fmaxf(v, dpp_mov_f32<0x114, 0xf>(ninf, v));
    v = fmaxf(v, dpp_mov_f32<0x118, 0xf>(ninf, v));
    v = fmaxf(v, dpp_mov_f32<0x142, 0xa>(ninf, v));
    v = fmaxf(v, dpp_mov_f32<0x143, 0xc>(ninf, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

}  // namespace detr
