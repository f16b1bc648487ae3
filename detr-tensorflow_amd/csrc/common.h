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
void launch_splitk_reduce(const float *ws, int splits, long long part_stride, int rows, int cols, float *C, long long ldc,
                          float alpha, const float *scale, hipStream_t stream);

// tuning hook: integer environment variable (0 when unset)
static inline int env_tile(const char *name) {
    const char *v = getenv(name);
    return v ? atoi(v) : 0;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Counter-based dropout: element `idx` of dropout site `seed` is kept iff the top 24 bits of a 32-bit
// integer hash are >= thresh16 = p * 2^16.  Forward and backward kernels regenerate the same mask from
// (seed, idx); tests/oracle restate the same hash in numpy (Keras Dropout semantics: kept values are
// scaled by 1/(1-p); the reference's TF RNG stream itself is not reproducible).
// One 32-bit hash serves the element PAIR (idx & ~1, idx | 1): the low / high 16 bits are compared with
// thresh16 = p * 2^16 (p = 0.1 -> 6553/65536).  Kernels whose lanes own adjacent elements (attention
// probabilities, float4 epilogues) therefore evaluate one hash per two elements.
__host__ __device__ __forceinline__ uint32_t drop_hash(uint32_t seed, unsigned long long pair) {
    uint32_t x = (uint32_t)pair ^ seed;
    x ^= (uint32_t)(pair >> 32) * 0x9E3779B9u;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ bool drop_keep(uint32_t seed, unsigned long long idx, uint32_t thresh16) {
    const uint32_t h = drop_hash(seed, idx >> 1);
    return ((idx & 1ull) ? (h >> 16) : (h & 0xFFFFu)) >= thresh16;
}
static inline uint32_t drop_thresh16(float p) { return (uint32_t)(p * 65536.0f); }

// wave64 reductions (all 64 lanes participate)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace detr
