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
extern "C" int detr_hip_memset_zero(void *ptr, size_t bytes, void *stream) {
    if (bytes == 0) return 0;
    hipError_t e = hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) {
        detr::set_error("memset_zero: %s", hipGetErrorString(e));
        return -2;
    }
    return 0;
}
