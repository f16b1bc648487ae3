// optim.hip -- multi-tensor optimiser step on a flat fp32 parameter buffer:
// per-TENSOR clip-by-norm (Keras `clipnorm`, i.e. tf.clip_by_norm) followed by Keras Adam
// (detr_tf/optimizers.py:86-88,137-163).  One launch computes every tensor's sum of squares,
// one launch applies clip + Adam to all tensors; HBM-bound (reads g,m,v,p; writes m,v,p).
#include "common.h"

namespace detr {

__global__ __launch_bounds__(256) void sumsq_segments_kernel(const float *__restrict__ g,
                                                             const int *__restrict__ chunk_tensor,
                                                             const long long *__restrict__ chunk_start,
                                                             const long long *__restrict__ seg_end, int chunk,
                                                             float *__restrict__ sumsq) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const int t = chunk_tensor[c];
    const long long s0 = chunk_start[c];
    long long s1 = s0 + chunk;
    if (s1 > seg_end[t]) s1 = seg_end[t];
    float acc = 0.f;
    for (long long i = s0 + threadIdx.x; i < s1; i += blockDim.x) {
        const float v = g[i];
        acc += v * v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    // one partial per chunk, plain store: the per-tensor sum is formed in a FIXED order by clip_adam_kernel, so that every
    // data-parallel replica (and every replay) clips with bit-identical norms -- fp32 atomics here made replicas drift by ulps
    if (threadIdx.x == 0) sumsq[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float *__restrict__ param, const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v,
                                                        const int *__restrict__ chunk_tensor,
                                                        const long long *__restrict__ chunk_start,
                                                        const long long *__restrict__ seg_end,
                                                        const int *__restrict__ tensor_group,
                                                        const float *__restrict__ sumsq,
                                                        const float *__restrict__ hyper, int chunk, int n_chunks) {
    const int c = blockIdx.x;
    const int t = chunk_tensor[c];
    const long long s0 = chunk_start[c];
    long long s1 = s0 + chunk;
    if (s1 > seg_end[t]) s1 = seg_end[t];
    const int grp = tensor_group[t];
    if (grp < 0) return;                       // group not trained this step (config.train_<group> false)
    const float lr_t = hyper[grp];
    const float clip = hyper[3], b1 = hyper[4], b2 = hyper[5], eps = hyper[6];
    // tf.clip_by_norm: g * clip / max(||g||, clip)   (clip <= 0 disables)
    float cs = 1.0f;
    if (clip > 0.f) {
        // the chunks of tensor t are consecutive: [lo, hi) by binary search, partials summed in a fixed order
        __shared__ int s_lo, s_hi;
        __shared__ float s_red[4];
        if (threadIdx.x == 0) {
            int a = 0, b = c;                      // first index with chunk_tensor >= t  (chunk_tensor[c] == t)
            while (a < b) { const int mid = (a + b) >> 1; if (chunk_tensor[mid] < t) a = mid + 1; else b = mid; }
            s_lo = a;
            a = c; b = n_chunks;                   // first index with chunk_tensor > t
            while (a < b) { const int mid = (a + b) >> 1; if (chunk_tensor[mid] <= t) a = mid + 1; else b = mid; }
            s_hi = a;
        }
        __syncthreads();
        float acc = 0.f;
        for (int k = s_lo + (int)threadIdx.x; k < s_hi; k += blockDim.x) acc += sumsq[k];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
        __syncthreads();
        const float nrm = sqrtf((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
        cs = clip / fmaxf(nrm, clip);
    }
    for (long long i = s0 + threadIdx.x; i < s1; i += blockDim.x) {
        const float gi = g[i] * cs;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        param[i] = param[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

__global__ void set_floats8_kernel(float *dst, float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                   float v7) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3; dst[4] = v4; dst[5] = v5; dst[6] = v6; dst[7] = v7;
    }
}

}  // namespace detr

using namespace detr;

extern "C" int detr_hip_set_floats8_f32(float *dst, float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                        float v7, void *stream) {
    DETR_REQUIRE(dst, "set_floats8: null dst");
    hipLaunchKernelGGL(set_floats8_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, v0, v1, v2, v3, v4, v5, v6, v7);
    DETR_LAUNCH_CHECK("set_floats8");
    return 0;
}

extern "C" int detr_hip_sumsq_segments_f32(const float *g, const int32_t *chunk_tensor, const int64_t *chunk_start,
                                           const int64_t *seg_end, int32_t n_chunks, int32_t chunk, float *sumsq,
                                           void *stream) {
    DETR_REQUIRE(g && chunk_tensor && chunk_start && seg_end && sumsq && n_chunks > 0 && chunk > 0, "sumsq: bad args");
    hipLaunchKernelGGL(sumsq_segments_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, g, chunk_tensor,
                       reinterpret_cast<const long long *>(chunk_start), reinterpret_cast<const long long *>(seg_end),
                       chunk, sumsq);
    DETR_LAUNCH_CHECK("sumsq_segments");
    return 0;
}

extern "C" int detr_hip_clip_adam_f32(float *param, const float *g, float *m, float *v, const int32_t *chunk_tensor,
                                      const int64_t *chunk_start, const int64_t *seg_end, const int32_t *tensor_group,
                                      const float *sumsq, const float *hyper, int32_t n_chunks, int32_t chunk,
                                      void *stream) {
    DETR_REQUIRE(param && g && m && v && chunk_tensor && chunk_start && seg_end && tensor_group && sumsq && hyper,
                 "clip_adam: null operand");
    DETR_REQUIRE(n_chunks > 0 && chunk > 0, "clip_adam: bad chunking");
    hipLaunchKernelGGL(clip_adam_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, param, g, m, v, chunk_tensor,
                       reinterpret_cast<const long long *>(chunk_start), reinterpret_cast<const long long *>(seg_end),
                       tensor_group, sumsq, hyper, chunk, n_chunks);
    DETR_LAUNCH_CHECK("clip_adam");
    return 0;
}
