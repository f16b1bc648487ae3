// common.hip -- error reporting and trivial entry points of libdetr_hip.so.
#include "common.h"
#include <string.h>

namespace detr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace detr

extern "C" const char *detr_hip_last_error(void) { return detr::g_err; }
extern "C" int detr_hip_abi_version(void) { return DETR_HIP_ABI_VERSION; }
// Zero fill as an ordinary KERNEL, not hipMemsetAsync: a memset node captured into a hipGraph (eval-forward / train-step
// replay) was observed to fill with stale non-zero patterns after other work had run between replays (the decoder's zero
// target read 1e-28 .. 5e-9 instead of 0 on the replay that followed an eager training step; a kernel node replays exactly).
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4 *__restrict__ p16, size_t n16, unsigned char *__restrict__ tail, int ntail) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p16[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

extern "C" int detr_hip_memset_zero(void *ptr, size_t bytes, void *stream) {
    if (bytes == 0) return 0;
    unsigned char *b = reinterpret_cast<unsigned char *>(ptr);
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(b) & 15)) & 15;          // bytes up to the first 16-byte boundary
    if (head >= bytes || bytes < 64) {                                                // tiny or unaligned-only: byte stores
        if (bytes > 255) { detr::set_error("memset_zero: %zu unaligned bytes", bytes); return -2; }
        hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nullptr, (size_t)0, b, (int)bytes);
    } else {
        if (head) hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nullptr, (size_t)0, b, (int)head);
        const size_t n16 = (bytes - head) / 16;
        const int ntail = (int)((bytes - head) % 16);
        size_t grid = (n16 + 255) / 256;
        if (grid > 2048) grid = 2048;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint4 *>(b + head), n16,
                           b + head + n16 * 16, ntail);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        detr::set_error("memset_zero: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}
