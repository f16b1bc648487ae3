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
