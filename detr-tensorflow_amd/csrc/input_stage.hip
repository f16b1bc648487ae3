// input_stage.hip -- device-side input stage: what the reference does on the host for every batch
// before the step can start --
//   * fixed-size resize of the uint8 image (data/transformation.py:82-91: imgaug Resize -> cv2.resize, third-party),
//   * normalisation to float32 (data/processing.py:6-21: "torch_resnet" (x/255 - mean)/std on RGB, "tf_resnet" BGR - mean),
//   * the padded target layout with its in-band header row (data/processing.py:35-55)
// -- as two launches on the step's stream, so the batch arrives in HBM as uint8 (a quarter of the PCIe bytes of the
// fp32 tensor) and never exists as a host float array.
//
// Normalisation is a 256 x 3 lookup table built by the host in float64 exactly as the reference's NumPy expression and
// rounded once to float32: bit-identical to `normalized_images` for every possible pixel value.  The resize is the
// documented cubic / linear / nearest kernel of cv2.resize in fp32 (half-pixel centres, replicated border, a = -0.75,
// result rounded half-to-even and saturated to uint8 before the lookup): cv2 / imgaug are not installable here, so that
// part is restated from their documentation ("parity unpinned"; identity when source and target sizes agree).
#include "common.h"

namespace detr {

struct InputArgs {
    const unsigned char *src; long long s_b;     // [B, Hs, Ws, 3] uint8, batch stride in bytes
    float *dst;                                  // [B, Hd, Wd, 3]
    const float *lut;                            // [3][256]: dst channel c = lut[c][resized src channel perm[c]]
    int B, Hs, Ws, Hd, Wd, interp;               // interp: 0 nearest, 1 linear, 2 cubic
    int perm0, perm1, perm2;
};

__device__ __forceinline__ void cubic_coeffs(float f, float (&c)[4]) {
    const float A = -0.75f;
    c[0] = ((A * (f + 1.0f) - 5.0f * A) * (f + 1.0f) + 8.0f * A) * (f + 1.0f) - 4.0f * A;
    c[1] = ((A + 2.0f) * f - (A + 3.0f)) * f * f + 1.0f;
    c[2] = ((A + 2.0f) * (1.0f - f) - (A + 3.0f)) * (1.0f - f) * (1.0f - f) + 1.0f;
    c[3] = 1.0f - c[0] - c[1] - c[2];
}

__global__ __launch_bounds__(256) void input_stage_kernel(InputArgs a) {
    const long long total = (long long)a.B * a.Hd * a.Wd;
    const float sy = (float)a.Hs / (float)a.Hd, sx = (float)a.Ws / (float)a.Wd;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.Wd);
        const long long t = i / a.Wd;
        const int y = (int)(t % a.Hd), b = (int)(t / a.Hd);
        const unsigned char *img = a.src + b * a.s_b;
        float v[3];
        if (a.Hs == a.Hd && a.Ws == a.Wd) {
            const unsigned char *p = img + ((long long)y * a.Ws + x) * 3;
            v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        } else {
            const float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
            if (a.interp == 0) {
                const int yy = min(max((int)floorf((float)y * sy), 0), a.Hs - 1), xx = min(max((int)floorf((float)x * sx), 0), a.Ws - 1);
                const unsigned char *p = img + ((long long)yy * a.Ws + xx) * 3;
                v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
            } else {
                const int n = a.interp == 1 ? 2 : 4;
                const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
                float cy[4], cx[4];
                if (a.interp == 1) {
                    cy[0] = 1.0f - (fy - (float)y0); cy[1] = fy - (float)y0;
                    cx[0] = 1.0f - (fx - (float)x0); cx[1] = fx - (float)x0;
                } else {
                    cubic_coeffs(fy - (float)y0, cy);
                    cubic_coeffs(fx - (float)x0, cx);
                }
                const int off = a.interp == 1 ? 0 : -1;
                float acc[3] = {0.f, 0.f, 0.f};
                for (int j = 0; j < n; ++j) {
                    const int yy = min(max(y0 + off + j, 0), a.Hs - 1);
                    float row[3] = {0.f, 0.f, 0.f};
                    for (int k = 0; k < n; ++k) {
                        const int xx = min(max(x0 + off + k, 0), a.Ws - 1);
                        const unsigned char *p = img + ((long long)yy * a.Ws + xx) * 3;
                        row[0] += cx[k] * (float)p[0]; row[1] += cx[k] * (float)p[1]; row[2] += cx[k] * (float)p[2];
                    }
                    acc[0] += cy[j] * row[0]; acc[1] += cy[j] * row[1]; acc[2] += cy[j] * row[2];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = fminf(fmaxf(rintf(acc[c]), 0.0f), 255.0f);      // back to uint8 like cv2 (saturate_cast)
            }
        }
        const int u0 = (int)v[a.perm0], u1 = (int)v[a.perm1], u2 = (int)v[a.perm2];
        float *o = a.dst + i * 3;
        o[0] = a.lut[u0]; o[1] = a.lut[256 + u1]; o[2] = a.lut[512 + u2];
    }
}

// targets: ragged (boxes [N,4], classes [N], offsets [B+1]) -> t_bbox [B,R,4] with header row [n,0,0,0], t_class [B,R] with
// header 0, zero padded (processing.py:35-55)
__global__ __launch_bounds__(128) void pad_labels_kernel(const float *__restrict__ boxes, const long long *__restrict__ classes,
                                                         const int *__restrict__ offsets, int R, float *__restrict__ t_bbox,
                                                         long long *__restrict__ t_class) {
    const int b = blockIdx.x;
    const int lo = offsets[b], n = offsets[b + 1] - lo;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        long long c = 0;
        if (r == 0) bx.x = (float)n;
        else if (r - 1 < n) {
            bx = reinterpret_cast<const float4 *>(boxes)[lo + r - 1];
            c = classes[lo + r - 1];
        }
        reinterpret_cast<float4 *>(t_bbox)[(long long)b * R + r] = bx;
        t_class[(long long)b * R + r] = c;
    }
}

}  // namespace detr

using namespace detr;

extern "C" int detr_hip_input_stage(const detr_input_desc *d, void *stream) {
    DETR_REQUIRE(d && d->src && d->dst && d->lut, "input_stage: null operand");
    DETR_REQUIRE(d->B > 0 && d->Hs > 0 && d->Ws > 0 && d->Hd > 0 && d->Wd > 0, "input_stage: bad shape");
    DETR_REQUIRE(d->interpolation >= 0 && d->interpolation <= 2, "input_stage: interpolation must be 0 (nearest), 1 (linear) or 2 (cubic)");
    for (int c = 0; c < 3; ++c) DETR_REQUIRE(d->perm[c] >= 0 && d->perm[c] < 3, "input_stage: channel permutation out of range");
    InputArgs a;
    a.src = d->src; a.s_b = d->src_batch_stride; a.dst = d->dst; a.lut = d->lut;
    a.B = d->B; a.Hs = d->Hs; a.Ws = d->Ws; a.Hd = d->Hd; a.Wd = d->Wd; a.interp = d->interpolation;
    a.perm0 = d->perm[0]; a.perm1 = d->perm[1]; a.perm2 = d->perm[2];
    const long long total = (long long)d->B * d->Hd * d->Wd;
    const int grid = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(input_stage_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    DETR_LAUNCH_CHECK("input_stage");
    return 0;
}

extern "C" int detr_hip_pad_labels(const float *boxes, const int64_t *classes, const int32_t *offsets, int32_t B, int32_t R,
                                   float *t_bbox, int64_t *t_class, void *stream) {
    DETR_REQUIRE(offsets && t_bbox && t_class && B > 0 && R > 1, "pad_labels: bad args");
    DETR_REQUIRE(aligned16(boxes) && aligned16(t_bbox), "pad_labels: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(pad_labels_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, boxes, reinterpret_cast<const long long *>(classes),
                       offsets, R, t_bbox, reinterpret_cast<long long *>(t_class));
    DETR_LAUNCH_CHECK("pad_labels");
    return 0;
}
