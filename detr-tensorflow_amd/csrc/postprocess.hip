// postprocess.hip -- batched post-processing of the DETR outputs on the device.
//
// The reference's get_model_inference (detr_tf/inference.py:68-95) handles batch element 0 only and runs five TF ops plus
// a tf.where / three gathers per call; its validation loop (logger/training_logging.py:61-88, eval.py:41-55) calls it
// once per image.  Here ONE launch serves the whole batch: per image one workgroup computes, per query, the softmax row
// statistics (score = max probability, label = first arg-max, inference.py:73-75), drops the background queries
// (:78-83, order preserved like tf.where) by an in-workgroup prefix count, converts the boxes (:86-93, bbox.py:171-196
// incl. the clip to [0, 1]) and writes the kept detections compacted per image.
#include "common.h"

namespace detr {

constexpr int PP_MAXQ = 1024;

struct PostArgs {
    const float *logits; long long sl_b, sl_q;      // [B, Q, C]
    const float *boxes; long long sb_b, sb_q;       // [B, Q, 4] cx, cy, w, h
    int B, Q, C, background, fmt;                   // fmt: 0 xy_center, 1 xyxy (clipped), 2 yxyx (clipped)
    float *out_boxes;                               // [B, Q, 4] compacted per image
    long long *out_labels;                          // [B, Q]
    float *out_scores;                              // [B, Q]
    int *counts;                                    // [B]
};

__global__ __launch_bounds__(256) void postprocess_kernel(PostArgs a) {
    __shared__ float s_score[PP_MAXQ];
    __shared__ int s_label[PP_MAXQ];
    __shared__ int s_wave_cnt[4];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *lg = a.logits + b * a.sl_b;
    for (int q = wave; q < a.Q; q += 4) {
        const float *row = lg + q * a.sl_q;
        float mx = -INFINITY;
        for (int c = lane; c < a.C; c += 64) mx = fmaxf(mx, row[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int c = lane; c < a.C; c += 64) s += expf(row[c] - mx);
        s = wave_sum(s);
        // arg-max of the PROBABILITIES (inference.py:75 takes argmax(softmax)), first occurrence on ties
        float best = -1.f;
        int arg = INT_MAX;
        for (int c = lane; c < a.C; c += 64) {
            const float p = expf(row[c] - mx) / s;
            if (p > best) { best = p; arg = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        if (lane == 0) { s_score[q] = best; s_label[q] = arg; }
    }
    __syncthreads();
    // order-preserving compaction: 256 queries per pass
    int base = 0;
    for (int q0 = 0; q0 < a.Q; q0 += 256) {
        const int q = q0 + threadIdx.x;
        const bool keep = q < a.Q && s_label[q] != a.background;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += s_wave_cnt[w];
        const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
        if (keep) {
            const float *bx = a.boxes + b * a.sb_b + q * a.sb_q;
            const float cx = bx[0], cy = bx[1], w = bx[2], h = bx[3];
            float o0 = cx, o1 = cy, o2 = w, o3 = h;
            if (a.fmt != 0) {
                const float x0 = fminf(fmaxf(cx - w / 2, 0.f), 1.f), y0 = fminf(fmaxf(cy - h / 2, 0.f), 1.f);
                const float x1 = fminf(fmaxf(cx + w / 2, 0.f), 1.f), y1 = fminf(fmaxf(cy + h / 2, 0.f), 1.f);
                if (a.fmt == 1) { o0 = x0; o1 = y0; o2 = x1; o3 = y1; }
                else { o0 = y0; o1 = x0; o2 = y1; o3 = x1; }
            }
            float *ob = a.out_boxes + ((long long)b * a.Q + pos) * 4;
            ob[0] = o0; ob[1] = o1; ob[2] = o2; ob[3] = o3;
            a.out_labels[(long long)b * a.Q + pos] = s_label[q];
            a.out_scores[(long long)b * a.Q + pos] = s_score[q];
        }
        base += s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.counts[b] = base;
}

}  // namespace detr

using namespace detr;

extern "C" int detr_hip_postprocess(const detr_postprocess_desc *d, void *stream) {
    DETR_REQUIRE(d && d->logits && d->boxes && d->out_boxes && d->out_labels && d->out_scores && d->counts, "postprocess: null operand");
    DETR_REQUIRE(d->B > 0 && d->Q > 0 && d->Q <= PP_MAXQ && d->C > 0, "postprocess: bad shape B=%d Q=%d C=%d (Q <= %d)", d->B, d->Q, d->C, PP_MAXQ);
    DETR_REQUIRE(d->bbox_format >= 0 && d->bbox_format <= 2, "postprocess: bbox_format must be 0 (xy_center), 1 (xyxy) or 2 (yxyx)");
    PostArgs a;
    a.logits = d->logits; a.sl_b = d->sL_b; a.sl_q = d->sL_q;
    a.boxes = d->boxes; a.sb_b = d->sB_b; a.sb_q = d->sB_q;
    a.B = d->B; a.Q = d->Q; a.C = d->C; a.background = d->background_class; a.fmt = d->bbox_format;
    a.out_boxes = d->out_boxes; a.out_labels = reinterpret_cast<long long *>(d->out_labels); a.out_scores = d->out_scores; a.counts = d->counts;
    hipLaunchKernelGGL(postprocess_kernel, dim3(d->B), dim3(256), 0, (hipStream_t)stream, a);
    DETR_LAUNCH_CHECK("postprocess");
    return 0;
}
