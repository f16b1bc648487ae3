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
static const char *const g_tune_names[T_COUNT] = {
    "DETR_HIP_GEMM_TILE",
    "DETR_HIP_SPLIT_XCD",
    "DETR_HIP_GEMM_STREAM",
    "DETR_HIP_GEMM_GROUP",
    "DETR_HIP_STREAM_SL",
    "DETR_HIP_STREAM_NW",
    "DETR_HIP_CONV_HALO",
    "DETR_HIP_CONV_TILE",
    "DETR_HIP_DGRAD_S2_CLASSES",
    "DETR_HIP_WGRAD_FUSED",
    "DETR_HIP_WGRAD_FUSED_WGS",
    "DETR_HIP_WGRAD_TILE",
    "DETR_HIP_STEM_ROWS",
    "DETR_HIP_ATTN_WAVES",
    "DETR_HIP_ATTN_SPLIT",
    "DETR_HIP_GEMM_K64",
    "DETR_HIP_EPI_WIDE",
    "DETR_HIP_SLAB_TS",
    "DETR_HIP_GEMM_RING",
    "DETR_HIP_RING_NS",
    "DETR_HIP_RING_BN",
    "DETR_HIP_RING_WGS",
    "DETR_HIP_RING_ROWS",
    "DETR_HIP_RING_WTILE",
    "DETR_HIP_RING_ABLATE",
    "DETR_HIP_CONV_DMA",
    "DETR_HIP_SPLIT3_T128",
    "DETR_HIP_X3_DB",
    "DETR_HIP_X3_T192",
    "DETR_HIP_X3_WG_ROUNDS",
    "DETR_HIP_X3_CONV",
    "DETR_HIP_SPLIT3_ALL",
};
static int g_tune[T_COUNT];
static void load_tuning() {
    for (int i = 0; i < T_COUNT; ++i) {
        const char *v = getenv(g_tune_names[i]);
        __atomic_store_n(&g_tune[i], v ? atoi(v) : 0, __ATOMIC_RELAXED);
    }
}
static const bool g_tune_loaded = (load_tuning(), true);      // at library load (dlopen), before any entry point can run
int tune(TuneKey k) { return __atomic_load_n(&g_tune[k], __ATOMIC_RELAXED); }
}  // namespace detr

extern "C" int detr_hip_reload_tuning(void) {
    detr::load_tuning();
    return 0;
}
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
// ---- ABI layout self-check ---------------------------------------------------------------------------------------------
// A foreign binder mirrors the descriptor structs of include/detr_hip.h by hand (ctypes / cgo / JNI).  This entry point
// reports what THIS build of the library sees: for struct `which`, out[0] = sizeof and out[1..] = offsetof of every field in
// declaration order.  The binding compares them with its own mirror at load time (detr_tf/_hip.py: check_struct_layouts).
#include <stddef.h>
#define DETR_OFF(S, f) (int32_t)offsetof(S, f)
extern "C" int detr_hip_struct_layout(int32_t which, int32_t *out, int32_t cap) {
    int n = 0;
#define DETR_PUT(v) do { if (n < cap) out[n] = (v); ++n; } while (0)
    switch (which) {
    case 0:   // detr_reduce_desc
        DETR_PUT((int32_t)sizeof(detr_reduce_desc));
        DETR_PUT(DETR_OFF(detr_reduce_desc, ws)); DETR_PUT(DETR_OFF(detr_reduce_desc, splits)); DETR_PUT(DETR_OFF(detr_reduce_desc, part_stride));
        DETR_PUT(DETR_OFF(detr_reduce_desc, rows)); DETR_PUT(DETR_OFF(detr_reduce_desc, cols)); DETR_PUT(DETR_OFF(detr_reduce_desc, C));
        DETR_PUT(DETR_OFF(detr_reduce_desc, ldc)); DETR_PUT(DETR_OFF(detr_reduce_desc, alpha)); DETR_PUT(DETR_OFF(detr_reduce_desc, scale));
        DETR_PUT(DETR_OFF(detr_reduce_desc, rs_ws)); DETR_PUT(DETR_OFF(detr_reduce_desc, rs_out)); DETR_PUT(DETR_OFF(detr_reduce_desc, rs_alpha));
        DETR_PUT(DETR_OFF(detr_reduce_desc, ts_bm)); DETR_PUT(DETR_OFF(detr_reduce_desc, ts_bn)); DETR_PUT(DETR_OFF(detr_reduce_desc, ts_tiles_n));
        break;
    case 1:   // detr_gemm_desc
        DETR_PUT((int32_t)sizeof(detr_gemm_desc));
        DETR_PUT(DETR_OFF(detr_gemm_desc, M)); DETR_PUT(DETR_OFF(detr_gemm_desc, N)); DETR_PUT(DETR_OFF(detr_gemm_desc, K));
        DETR_PUT(DETR_OFF(detr_gemm_desc, A)); DETR_PUT(DETR_OFF(detr_gemm_desc, lda)); DETR_PUT(DETR_OFF(detr_gemm_desc, a_kcontig));
        DETR_PUT(DETR_OFF(detr_gemm_desc, B)); DETR_PUT(DETR_OFF(detr_gemm_desc, ldb)); DETR_PUT(DETR_OFF(detr_gemm_desc, b_kcontig));
        DETR_PUT(DETR_OFF(detr_gemm_desc, C)); DETR_PUT(DETR_OFF(detr_gemm_desc, ldc)); DETR_PUT(DETR_OFF(detr_gemm_desc, batch));
        DETR_PUT(DETR_OFF(detr_gemm_desc, batch_inner)); DETR_PUT(DETR_OFF(detr_gemm_desc, sA0)); DETR_PUT(DETR_OFF(detr_gemm_desc, sA1));
        DETR_PUT(DETR_OFF(detr_gemm_desc, sB0)); DETR_PUT(DETR_OFF(detr_gemm_desc, sB1)); DETR_PUT(DETR_OFF(detr_gemm_desc, sC0));
        DETR_PUT(DETR_OFF(detr_gemm_desc, sC1)); DETR_PUT(DETR_OFF(detr_gemm_desc, alpha)); DETR_PUT(DETR_OFF(detr_gemm_desc, scale));
        DETR_PUT(DETR_OFF(detr_gemm_desc, bias)); DETR_PUT(DETR_OFF(detr_gemm_desc, residual)); DETR_PUT(DETR_OFF(detr_gemm_desc, ldr));
        DETR_PUT(DETR_OFF(detr_gemm_desc, mask)); DETR_PUT(DETR_OFF(detr_gemm_desc, ldmask)); DETR_PUT(DETR_OFF(detr_gemm_desc, act));
        DETR_PUT(DETR_OFF(detr_gemm_desc, split_k)); DETR_PUT(DETR_OFF(detr_gemm_desc, workspace)); DETR_PUT(DETR_OFF(detr_gemm_desc, workspace_bytes));
        DETR_PUT(DETR_OFF(detr_gemm_desc, dropout_p)); DETR_PUT(DETR_OFF(detr_gemm_desc, dropout_seed)); DETR_PUT(DETR_OFF(detr_gemm_desc, compute));
        DETR_PUT(DETR_OFF(detr_gemm_desc, rowsum_a)); DETR_PUT(DETR_OFF(detr_gemm_desc, rowsum_alpha)); DETR_PUT(DETR_OFF(detr_gemm_desc, b_dtype));
        DETR_PUT(DETR_OFF(detr_gemm_desc, a_dtype)); DETR_PUT(DETR_OFF(detr_gemm_desc, c_dtype)); DETR_PUT(DETR_OFF(detr_gemm_desc, r_dtype));
        DETR_PUT(DETR_OFF(detr_gemm_desc, m_dtype)); DETR_PUT(DETR_OFF(detr_gemm_desc, dropout_step)); DETR_PUT(DETR_OFF(detr_gemm_desc, defer_out));
        DETR_PUT(DETR_OFF(detr_gemm_desc, maskbits_out)); DETR_PUT(DETR_OFF(detr_gemm_desc, ld_maskbits_out));
        DETR_PUT(DETR_OFF(detr_gemm_desc, ln_gamma)); DETR_PUT(DETR_OFF(detr_gemm_desc, ln_beta)); DETR_PUT(DETR_OFF(detr_gemm_desc, ln_y));
        DETR_PUT(DETR_OFF(detr_gemm_desc, ln_mean)); DETR_PUT(DETR_OFF(detr_gemm_desc, ln_rstd)); DETR_PUT(DETR_OFF(detr_gemm_desc, ln_add));
        DETR_PUT(DETR_OFF(detr_gemm_desc, ln_add_rows)); DETR_PUT(DETR_OFF(detr_gemm_desc, ln_y2)); DETR_PUT(DETR_OFF(detr_gemm_desc, ln_y16));
        DETR_PUT(DETR_OFF(detr_gemm_desc, ln_eps));
        break;
    case 2:   // detr_conv3x3_desc
        DETR_PUT((int32_t)sizeof(detr_conv3x3_desc));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, N)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, Hi)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, Wi));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, Ci)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, Ho)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, Wo));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, Co)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, stride)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, pad));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, x)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, w)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, y));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, alpha)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, scale)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, bias));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, residual)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, mask)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, act));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, split)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, workspace)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, workspace_bytes));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, compute)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, w_dtype)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, x_dtype));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, y_dtype)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, r_dtype)); DETR_PUT(DETR_OFF(detr_conv3x3_desc, m_dtype));
        DETR_PUT(DETR_OFF(detr_conv3x3_desc, maskbits_out));
        break;
    case 3:   // detr_stem_desc
        DETR_PUT((int32_t)sizeof(detr_stem_desc));
        DETR_PUT(DETR_OFF(detr_stem_desc, N)); DETR_PUT(DETR_OFF(detr_stem_desc, H)); DETR_PUT(DETR_OFF(detr_stem_desc, W));
        DETR_PUT(DETR_OFF(detr_stem_desc, Ho)); DETR_PUT(DETR_OFF(detr_stem_desc, Wo)); DETR_PUT(DETR_OFF(detr_stem_desc, img));
        DETR_PUT(DETR_OFF(detr_stem_desc, w)); DETR_PUT(DETR_OFF(detr_stem_desc, y)); DETR_PUT(DETR_OFF(detr_stem_desc, alpha));
        DETR_PUT(DETR_OFF(detr_stem_desc, scale)); DETR_PUT(DETR_OFF(detr_stem_desc, bias)); DETR_PUT(DETR_OFF(detr_stem_desc, act));
        DETR_PUT(DETR_OFF(detr_stem_desc, split)); DETR_PUT(DETR_OFF(detr_stem_desc, workspace)); DETR_PUT(DETR_OFF(detr_stem_desc, workspace_bytes));
        DETR_PUT(DETR_OFF(detr_stem_desc, compute)); DETR_PUT(DETR_OFF(detr_stem_desc, w_dtype)); DETR_PUT(DETR_OFF(detr_stem_desc, y_dtype));
        break;
    case 4:   // detr_layernorm_desc
        DETR_PUT((int32_t)sizeof(detr_layernorm_desc));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, rows)); DETR_PUT(DETR_OFF(detr_layernorm_desc, C)); DETR_PUT(DETR_OFF(detr_layernorm_desc, eps));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, x)); DETR_PUT(DETR_OFF(detr_layernorm_desc, gamma)); DETR_PUT(DETR_OFF(detr_layernorm_desc, beta));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, y)); DETR_PUT(DETR_OFF(detr_layernorm_desc, mean)); DETR_PUT(DETR_OFF(detr_layernorm_desc, rstd));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, add)); DETR_PUT(DETR_OFF(detr_layernorm_desc, add_rows)); DETR_PUT(DETR_OFF(detr_layernorm_desc, y2));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, y16)); DETR_PUT(DETR_OFF(detr_layernorm_desc, dy)); DETR_PUT(DETR_OFF(detr_layernorm_desc, dx));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, dgamma)); DETR_PUT(DETR_OFF(detr_layernorm_desc, dbeta)); DETR_PUT(DETR_OFF(detr_layernorm_desc, workspace));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, workspace_bytes)); DETR_PUT(DETR_OFF(detr_layernorm_desc, dx_add));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, dx_drop)); DETR_PUT(DETR_OFF(detr_layernorm_desc, dropout_p)); DETR_PUT(DETR_OFF(detr_layernorm_desc, dropout_site));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, dropout_step)); DETR_PUT(DETR_OFF(detr_layernorm_desc, dx_drop16));
        DETR_PUT(DETR_OFF(detr_layernorm_desc, defer_blocks_out)); DETR_PUT(DETR_OFF(detr_layernorm_desc, dy_add));
        break;
    case 5:   // detr_attn_desc
        DETR_PUT((int32_t)sizeof(detr_attn_desc));
        DETR_PUT(DETR_OFF(detr_attn_desc, B)); DETR_PUT(DETR_OFF(detr_attn_desc, H)); DETR_PUT(DETR_OFF(detr_attn_desc, T));
        DETR_PUT(DETR_OFF(detr_attn_desc, S)); DETR_PUT(DETR_OFF(detr_attn_desc, q)); DETR_PUT(DETR_OFF(detr_attn_desc, ldq));
        DETR_PUT(DETR_OFF(detr_attn_desc, k)); DETR_PUT(DETR_OFF(detr_attn_desc, ldk)); DETR_PUT(DETR_OFF(detr_attn_desc, v));
        DETR_PUT(DETR_OFF(detr_attn_desc, ldv)); DETR_PUT(DETR_OFF(detr_attn_desc, o)); DETR_PUT(DETR_OFF(detr_attn_desc, ldo));
        DETR_PUT(DETR_OFF(detr_attn_desc, lse)); DETR_PUT(DETR_OFF(detr_attn_desc, d_o)); DETR_PUT(DETR_OFF(detr_attn_desc, ldd_o));
        DETR_PUT(DETR_OFF(detr_attn_desc, dq)); DETR_PUT(DETR_OFF(detr_attn_desc, lddq)); DETR_PUT(DETR_OFF(detr_attn_desc, dk));
        DETR_PUT(DETR_OFF(detr_attn_desc, lddk)); DETR_PUT(DETR_OFF(detr_attn_desc, dv)); DETR_PUT(DETR_OFF(detr_attn_desc, lddv));
        DETR_PUT(DETR_OFF(detr_attn_desc, delta)); DETR_PUT(DETR_OFF(detr_attn_desc, scale)); DETR_PUT(DETR_OFF(detr_attn_desc, dropout_p));
        DETR_PUT(DETR_OFF(detr_attn_desc, dropout_site)); DETR_PUT(DETR_OFF(detr_attn_desc, dropout_step)); DETR_PUT(DETR_OFF(detr_attn_desc, compute));
        DETR_PUT(DETR_OFF(detr_attn_desc, io_dtype)); DETR_PUT(DETR_OFF(detr_attn_desc, dropmask));
        break;
    case 6:   // detr_setloss_desc
        DETR_PUT((int32_t)sizeof(detr_setloss_desc));
        DETR_PUT(DETR_OFF(detr_setloss_desc, levels)); DETR_PUT(DETR_OFF(detr_setloss_desc, B)); DETR_PUT(DETR_OFF(detr_setloss_desc, Q));
        DETR_PUT(DETR_OFF(detr_setloss_desc, C)); DETR_PUT(DETR_OFF(detr_setloss_desc, R)); DETR_PUT(DETR_OFF(detr_setloss_desc, logits));
        DETR_PUT(DETR_OFF(detr_setloss_desc, sL_l)); DETR_PUT(DETR_OFF(detr_setloss_desc, sL_b)); DETR_PUT(DETR_OFF(detr_setloss_desc, sL_q));
        DETR_PUT(DETR_OFF(detr_setloss_desc, boxes)); DETR_PUT(DETR_OFF(detr_setloss_desc, sB_l)); DETR_PUT(DETR_OFF(detr_setloss_desc, sB_b));
        DETR_PUT(DETR_OFF(detr_setloss_desc, sB_q)); DETR_PUT(DETR_OFF(detr_setloss_desc, t_bbox)); DETR_PUT(DETR_OFF(detr_setloss_desc, t_class));
        DETR_PUT(DETR_OFF(detr_setloss_desc, background_class));
        break;
    case 7:   // detr_input_desc
        DETR_PUT((int32_t)sizeof(detr_input_desc));
        DETR_PUT(DETR_OFF(detr_input_desc, B)); DETR_PUT(DETR_OFF(detr_input_desc, Hs)); DETR_PUT(DETR_OFF(detr_input_desc, Ws));
        DETR_PUT(DETR_OFF(detr_input_desc, Hd)); DETR_PUT(DETR_OFF(detr_input_desc, Wd)); DETR_PUT(DETR_OFF(detr_input_desc, src));
        DETR_PUT(DETR_OFF(detr_input_desc, src_batch_stride)); DETR_PUT(DETR_OFF(detr_input_desc, dst)); DETR_PUT(DETR_OFF(detr_input_desc, lut));
        DETR_PUT(DETR_OFF(detr_input_desc, perm)); DETR_PUT(DETR_OFF(detr_input_desc, interpolation));
        break;
    case 8:   // detr_postprocess_desc
        DETR_PUT((int32_t)sizeof(detr_postprocess_desc));
        DETR_PUT(DETR_OFF(detr_postprocess_desc, B)); DETR_PUT(DETR_OFF(detr_postprocess_desc, Q)); DETR_PUT(DETR_OFF(detr_postprocess_desc, C));
        DETR_PUT(DETR_OFF(detr_postprocess_desc, logits)); DETR_PUT(DETR_OFF(detr_postprocess_desc, sL_b)); DETR_PUT(DETR_OFF(detr_postprocess_desc, sL_q));
        DETR_PUT(DETR_OFF(detr_postprocess_desc, boxes)); DETR_PUT(DETR_OFF(detr_postprocess_desc, sB_b)); DETR_PUT(DETR_OFF(detr_postprocess_desc, sB_q));
        DETR_PUT(DETR_OFF(detr_postprocess_desc, background_class)); DETR_PUT(DETR_OFF(detr_postprocess_desc, bbox_format));
        DETR_PUT(DETR_OFF(detr_postprocess_desc, out_boxes)); DETR_PUT(DETR_OFF(detr_postprocess_desc, out_labels));
        DETR_PUT(DETR_OFF(detr_postprocess_desc, out_scores)); DETR_PUT(DETR_OFF(detr_postprocess_desc, counts));
        break;
    default:
        detr::set_error("struct_layout: unknown struct %d", which);
        return -1;
    }
#undef DETR_PUT
    return n;        // number of values (may exceed cap: call again with a larger buffer)
}
