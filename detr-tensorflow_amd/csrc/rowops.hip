// rowops.hip -- HBM-bound row / elementwise kernels of the transformer and the backbone glue
// (LayerNorm, attention softmax, bias-gradient column sums, broadcast adds, sigmoid/ReLU
// backward, frozen-BN folding).  One wave64 per row for the row kernels, float4 accesses,
// wave shuffles for the reductions.  Reference lines: see include/detr_hip.h.
#include "common.h"

namespace detr {

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim C (C % 4 == 0, C <= 1024): one wave per row, each lane holds
// up to 4 float4 (lane-strided so that a wave reads 1 KiB contiguous per step).
// ------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 4;

__device__ __forceinline__ unsigned ln_pk_bf16(float a, float b) {      // two RNE roundings (v_cvt_pk_bf16_f32)
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ y,
                                                            float *__restrict__ mean, float *__restrict__ rstd,
                                                            int rows, int C, float eps, const float *__restrict__ add,
                                                            int add_rows, float *__restrict__ y2, unsigned short *__restrict__ y16) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int nv = C >> 2;  // float4 per row
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
        const float4 *xr = reinterpret_cast<const float4 *>(x + (long long)row * C);
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int j = lane + 64 * i;
            v[i] = (j < nv) ? xr[j] : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int j = lane + 64 * i;
            if (j < nv) {
                const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
                q += (a * a + b * b) + (c * c + d * d);
            }
        }
        const float var = wave_sum(q) / (float)C;
        const float rs = rsqrtf(var + eps);
        float4 *yr = reinterpret_cast<float4 *>(y + (long long)row * C);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int j = lane + 64 * i;
            if (j < nv) {
                const float4 g = reinterpret_cast<const float4 *>(gamma)[j];
                const float4 b = reinterpret_cast<const float4 *>(beta)[j];
                float4 o;
                o.x = (v[i].x - mu) * rs * g.x + b.x;
                o.y = (v[i].y - mu) * rs * g.y + b.y;
                o.z = (v[i].z - mu) * rs * g.z + b.z;
                o.w = (v[i].w - mu) * rs * g.w + b.w;
                yr[j] = o;
                if (y16)         // bf16 twin of y: the A operand of the next bf16-compute GEMM (the loader would round it anyway)
                    reinterpret_cast<uint2 *>(y16 + (long long)row * C)[j] = make_uint2(ln_pk_bf16(o.x, o.y), ln_pk_bf16(o.z, o.w));
                if (y2) {        // the `+ pos` / `+ query_pos` operand of the next attention block, written while y is in registers
                    const float4 p = reinterpret_cast<const float4 *>(add + (long long)(row % add_rows) * C)[j];
                    reinterpret_cast<float4 *>(y2 + (long long)row * C)[j] = make_float4(o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w);
                }
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ; dgamma += sum dy*xhat ; dbeta += sum dy
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, float *__restrict__ dx,
                                                            float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                            int rows, int C, float *__restrict__ partial,
                                                            const float *__restrict__ dx_add, float *__restrict__ dx_drop,
                                                            float drop_scale, uint32_t drop_thresh, uint32_t drop_site,
                                                            const uint32_t *__restrict__ drop_step,
                                                            unsigned short *__restrict__ dx_drop16,
                                                            const float *__restrict__ dy_add) {
    __shared__ float red[2][4][64 * LN_MAXV * 4 / 4];  // [gamma|beta][wave][C] ; C <= 1024 -> see below
    // (partial sums are kept per lane in registers and reduced through LDS at the end)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int nv = C >> 2;
    float4 pg[LN_MAXV], pb[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        pg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool want_drop = dx_drop != nullptr || dx_drop16 != nullptr;
    const uint32_t dkey = (want_drop && drop_scale != 0.0f) ? drop_key(drop_site, drop_step) : 0u;
    for (int row = blockIdx.x * wpb + wave; row < rows; row += gridDim.x * wpb) {
        const float4 *xr = reinterpret_cast<const float4 *>(x + (long long)row * C);
        const float4 *dr = reinterpret_cast<const float4 *>(dy + (long long)row * C);
        const float mu = mean[row], rs = rstd[row];
        float4 xh[LN_MAXV], g[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int j = lane + 64 * i;
            if (j < nv) {
                const float4 xv = xr[j];
                float4 dv = dr[j];
                if (dy_add) {    // two gradient branches meet at this LayerNorm's output
                    const float4 e = reinterpret_cast<const float4 *>(dy_add + (long long)row * C)[j];
                    dv.x += e.x; dv.y += e.y; dv.z += e.z; dv.w += e.w;
                }
                const float4 gm = reinterpret_cast<const float4 *>(gamma)[j];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                pg[i].x += dv.x * xh[i].x; pg[i].y += dv.y * xh[i].y; pg[i].z += dv.z * xh[i].z; pg[i].w += dv.w * xh[i].w;
                pb[i].x += dv.x; pb[i].y += dv.y; pb[i].z += dv.z; pb[i].w += dv.w;
            } else {
                xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                g[i] = xh[i];
            }
        }
        const float m1 = wave_sum(s1) / (float)C;
        const float m2 = wave_sum(s2) / (float)C;
        float4 *oxr = reinterpret_cast<float4 *>(dx + (long long)row * C);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int j = lane + 64 * i;
            if (j < nv) {
                float4 o;
                o.x = rs * (g[i].x - m1 - xh[i].x * m2);
                o.y = rs * (g[i].y - m1 - xh[i].y * m2);
                o.z = rs * (g[i].z - m1 - xh[i].z * m2);
                o.w = rs * (g[i].w - m1 - xh[i].w * m2);
                if (dx_add) {    // a second gradient branch into the same tensor (e.g. the residual path of the next layer)
                    const float4 e = reinterpret_cast<const float4 *>(dx_add + (long long)row * C)[j];
                    o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
                }
                oxr[j] = o;
                if (want_drop) { // gradient through the dropout that precedes the residual add (same mask as the forward GEMM epilogue)
                    float4 d = o;
                    if (drop_scale != 0.0f) {
                        const unsigned long long di = (unsigned long long)row * C + 4 * j;      // even: two hashes serve the four elements
                        const uint32_t h0 = drop_hash(dkey, di >> 1), h1 = drop_hash(dkey, (di >> 1) + 1);
                        d.x = (h0 & 0xFFFFu) >= drop_thresh ? o.x * drop_scale : 0.0f;
                        d.y = (h0 >> 16) >= drop_thresh ? o.y * drop_scale : 0.0f;
                        d.z = (h1 & 0xFFFFu) >= drop_thresh ? o.z * drop_scale : 0.0f;
                        d.w = (h1 >> 16) >= drop_thresh ? o.w * drop_scale : 0.0f;
                    }
                    if (dx_drop) reinterpret_cast<float4 *>(dx_drop + (long long)row * C)[j] = d;
                    if (dx_drop16)   // bf16 twin (the gradient only feeds bf16-compute GEMM operands)
                        reinterpret_cast<uint2 *>(dx_drop16 + (long long)row * C)[j] = make_uint2(ln_pk_bf16(d.x, d.y), ln_pk_bf16(d.z, d.w));
                }
            }
        }
    }
    // block reduction of the per-wave column partials, then one atomic per column per block
    float *rg = &red[0][0][0];
    float *rb = &red[1][0][0];
    // layout red[which][wave][256] only holds C<=256 per pass: loop over the LN_MAXV chunks
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int j = lane + 64 * i;  // float4 index
        __syncthreads();
        rg[wave * 256 + lane * 4 + 0] = pg[i].x; rg[wave * 256 + lane * 4 + 1] = pg[i].y;
        rg[wave * 256 + lane * 4 + 2] = pg[i].z; rg[wave * 256 + lane * 4 + 3] = pg[i].w;
        rb[wave * 256 + lane * 4 + 0] = pb[i].x; rb[wave * 256 + lane * 4 + 1] = pb[i].y;
        rb[wave * 256 + lane * 4 + 2] = pb[i].z; rb[wave * 256 + lane * 4 + 3] = pb[i].w;
        __syncthreads();
        if (wave == 0 && j < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sg = 0.f, sb = 0.f;
                for (int w = 0; w < wpb; ++w) {
                    sg += rg[w * 256 + lane * 4 + e];
                    sb += rb[w * 256 + lane * 4 + e];
                }
                if (partial) {      // deterministic: per-block partials, summed in block order by the finish kernel
                    partial[(long long)blockIdx.x * 2 * C + j * 4 + e] = sg;
                    partial[(long long)blockIdx.x * 2 * C + C + j * 4 + e] = sb;
                } else {
                    unsafeAtomicAdd(dgamma + j * 4 + e, sg);
                    unsafeAtomicAdd(dbeta + j * 4 + e, sb);
                }
            }
        }
    }
}

// 64 columns x 16 row-partitions per block: every thread sums nblk/16 partials (independent loads), the partitions are
// combined through LDS in a fixed order
__global__ __launch_bounds__(1024) void layernorm_bwd_finish_kernel(const float *__restrict__ partial, int nblk, int C,
                                                                    float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < 2 * C)
        for (int b = part; b < nblk; b += 16) s += partial[(long long)b * 2 * C + c];
    red[part][lane] = s;
    __syncthreads();
    if (part == 0 && c < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        if (c < C) dgamma[c] += t;
        else dbeta[c - C] += t;
    }
}

// out[c] += alpha * sum_r x[r*ld + c] ; grid (col blocks of 256, row chunks)
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, float *__restrict__ out, long long rows,
                                                     int cols, long long ld, float alpha, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
    float s0 = 0.f, s1 = 0.f;
    long long r = r0;
    for (; r + 1 < r1; r += 2) {
        s0 += x[r * ld + c];
        s1 += x[(r + 1) * ld + c];
    }
    if (r < r1) s0 += x[r * ld + c];
    unsafeAtomicAdd(out + c, alpha * (s0 + s1));
}

__global__ void add_bcast_kernel(const float4 *__restrict__ x, const float4 *__restrict__ p, float4 *__restrict__ out,
                                 long long n4, long long period4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = x[i], b = p[i % period4];
        out[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

__global__ void add_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

__global__ void sigmoid_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, float *__restrict__ dz,
                                   long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = y[i];
        dz[i] = dy[i] * v * (1.0f - v);
    }
}

__global__ void dropout_kernel(const float *__restrict__ x, float *__restrict__ out, long long n, float scale,
                               uint32_t thresh, uint32_t site, const uint32_t *__restrict__ step) {
    const uint32_t seed = drop_key(site, step);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = drop_keep(seed, (unsigned long long)i, thresh) ? x[i] * scale : 0.0f;
}

__global__ void relu_mask_kernel(const float *__restrict__ g, const float *__restrict__ ref, float *__restrict__ out,
                                 long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = ref[i] > 0.0f ? g[i] : 0.0f;
}

__global__ void scale_cols_kernel(const float *__restrict__ w, const float *__restrict__ scale, float *__restrict__ o,
                                  long long n, int cols) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        o[i] = w[i] * scale[i % cols];
}

// o[c][r] = w[r][c] * scale[c]  (transposed scaled copy: K-contiguous B operand for the bf16 1x1-conv forward)
__global__ void scale_cols_t_kernel(const float *__restrict__ w, const float *__restrict__ scale, float *__restrict__ o,
                                    int rows, int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? w[(long long)r * cols + c] * (scale ? scale[c] : 1.0f) : 0.0f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) o[(long long)c * rows + r] = tile[tx][j];
    }
}

__global__ void bn_fold_kernel(const float *__restrict__ weight, const float *__restrict__ bias,
                               const float *__restrict__ mean, const float *__restrict__ var, float *__restrict__ scale,
                               float *__restrict__ shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float s = weight[c] * rsqrtf(var[c] + eps);
        scale[c] = s;
        shift[c] = bias[c] - mean[c] * s;
    }
}

__global__ void axpy_kernel(float *__restrict__ acc, const float *__restrict__ g, float a, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        acc[i] += a * g[i];
}

static inline int ew_grid(long long total, int block) {
    long long g = (total + block - 1) / block;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace detr

using namespace detr;

extern "C" int detr_hip_layernorm_fwd(const detr_layernorm_desc *d, void *stream) {
    DETR_REQUIRE(d && d->x && d->gamma && d->beta && d->y && d->mean && d->rstd, "layernorm fwd: null operand");
    const int C = d->C, rows = d->rows;
    DETR_REQUIRE(C % 4 == 0 && C <= 256 * LN_MAXV && rows > 0, "layernorm fwd: C=%d rows=%d unsupported", C, rows);
    DETR_REQUIRE(aligned16(d->x) && aligned16(d->y) && aligned16(d->gamma) && aligned16(d->beta), "layernorm fwd: alignment");
    if (d->y2) DETR_REQUIRE(d->add && d->add_rows > 0 && aligned16(d->add) && aligned16(d->y2), "layernorm fwd: y2 needs add / add_rows (16-byte aligned)");
    if (d->y16) DETR_REQUIRE((reinterpret_cast<uintptr_t>(d->y16) & 7) == 0, "layernorm fwd: y16 must be 8-byte aligned");
    const int grid = min(cdiv(rows, 4), 4096);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d->x, d->gamma, d->beta, d->y, d->mean,
                       d->rstd, rows, C, d->eps, d->add, d->add_rows, d->y2, d->y16);
    DETR_LAUNCH_CHECK("layernorm fwd");
    return 0;
}

extern "C" int64_t detr_hip_workspace_bytes_layernorm(const detr_layernorm_desc *d) {
    if (!d || d->rows <= 0 || d->C <= 0) return -1;
    return (int64_t)min(cdiv(d->rows, 8), 512) * 2 * d->C * 4;       // per-block gamma / beta partials of the backward
}

extern "C" int detr_hip_layernorm_bwd(const detr_layernorm_desc *d, void *stream) {
    DETR_REQUIRE(d && d->dy && d->x && d->gamma && d->mean && d->rstd && d->dx && d->dgamma && d->dbeta, "layernorm bwd: null operand");
    const int C = d->C, rows = d->rows;
    DETR_REQUIRE(C % 4 == 0 && C <= 256 * LN_MAXV && rows > 0, "layernorm bwd: C=%d rows=%d unsupported", C, rows);
    DETR_REQUIRE(aligned16(d->x) && aligned16(d->dy) && aligned16(d->dx) && aligned16(d->gamma), "layernorm bwd: alignment");
    if (d->dx_add) DETR_REQUIRE(aligned16(d->dx_add), "layernorm bwd: dx_add alignment");
    float drop_scale = 0.0f;
    if (d->dx_drop || d->dx_drop16) {
        DETR_REQUIRE((!d->dx_drop || aligned16(d->dx_drop)) && (reinterpret_cast<uintptr_t>(d->dx_drop16) & 7) == 0 &&
                     d->dropout_p >= 0.0f && d->dropout_p < 1.0f, "layernorm bwd: dx_drop needs 0 <= p < 1 and aligned outputs");
        drop_scale = d->dropout_p > 0.0f ? 1.0f / (1.0f - d->dropout_p) : 0.0f;      // p = 0: plain copies of dx
    }
    const int grid = min(cdiv(rows, 8), 512);
    // with a workspace of grid*2*C floats the gamma / beta gradients are reduced deterministically (per-block partials +
    // a finish launch); without one they are accumulated with fp32 atomics
    float *partial = (d->workspace && d->workspace_bytes >= (long long)grid * 2 * C * 4) ? d->workspace : nullptr;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d->dy, d->x, d->gamma, d->mean, d->rstd,
                       d->dx, d->dgamma, d->dbeta, rows, C, partial, d->dx_add, d->dx_drop, drop_scale, drop_thresh16(d->dropout_p),
                       d->dropout_site, d->dropout_step, d->dx_drop16, d->dy_add);
    DETR_LAUNCH_CHECK("layernorm bwd");
    if (d->defer_blocks_out) {
        DETR_REQUIRE(partial != nullptr, "layernorm bwd: defer_blocks_out needs a workspace of %d * 2 * C floats", grid);
        *d->defer_blocks_out = grid;
        return 0;
    }
    if (partial) {
        hipLaunchKernelGGL(layernorm_bwd_finish_kernel, dim3(cdiv(2 * C, 64)), dim3(1024), 0, (hipStream_t)stream, partial, grid,
                           C, d->dgamma, d->dbeta);
        DETR_LAUNCH_CHECK("layernorm bwd finish");
    }
    return 0;
}

namespace detr {
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk2_bf16(float a, float b) {
    bf16x2_t r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}
__global__ void cvt_bf16_kernel(const float4 *__restrict__ x, uint2 *__restrict__ out, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        out[i] = make_uint2(pk2_bf16(v.x, v.y), pk2_bf16(v.z, v.w));
    }
}
__global__ void scale_cols_bf16_kernel(const float4 *__restrict__ w, const float4 *__restrict__ scale, uint2 *__restrict__ out,
                                       long long n4, int c4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = w[i], sc = scale[i % c4];
        out[i] = make_uint2(pk2_bf16(v.x * sc.x, v.y * sc.y), pk2_bf16(v.z * sc.z, v.w * sc.w));
    }
}

// detr_tf/engine.py builds the table as int64[n][5] rows (w, scale, out, n4, c4 | reserved << 32)
static_assert(sizeof(detr_scale_entry) == 40, "detr_scale_entry layout changed: update Engine._refold_group");

__global__ void scale_cols_bf16_group_kernel(const detr_scale_entry *__restrict__ tab) {
    const detr_scale_entry e = tab[blockIdx.y];
    const float4 *w = reinterpret_cast<const float4 *>(e.w), *scale = reinterpret_cast<const float4 *>(e.scale);
    uint2 *out = reinterpret_cast<uint2 *>(e.out);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < e.n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = w[i], sc = scale[i % e.c4];
        out[i] = make_uint2(pk2_bf16(v.x * sc.x, v.y * sc.y), pk2_bf16(v.z * sc.z, v.w * sc.w));
    }
}
}  // namespace detr

extern "C" int detr_hip_scale_cols_bf16_group(const detr_scale_entry *table, int32_t n, void *stream) {
    DETR_REQUIRE(table && n > 0 && n <= 65535, "scale_cols_bf16_group: bad args");
    hipLaunchKernelGGL(scale_cols_bf16_group_kernel, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream, table);
    DETR_LAUNCH_CHECK("scale_cols_bf16_group");
    return 0;
}

extern "C" int detr_hip_cvt_bf16(const float *x, uint16_t *out, int64_t n, void *stream) {
    DETR_REQUIRE(x && out && n > 0 && n % 4 == 0 && aligned16(x) && ((uintptr_t)out % 8 == 0), "cvt_bf16: bad args");
    const long long n4 = n / 4;
    long long grid = (n4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float4 *)x, (uint2 *)out, n4);
    DETR_LAUNCH_CHECK("cvt_bf16");
    return 0;
}

extern "C" int detr_hip_scale_cols_bf16(const float *w, const float *scale, uint16_t *out, int64_t rows, int32_t cols,
                                        void *stream) {
    DETR_REQUIRE(w && scale && out && rows > 0 && cols > 0 && cols % 4 == 0, "scale_cols_bf16: bad args");
    DETR_REQUIRE(aligned16(w) && aligned16(scale) && ((uintptr_t)out % 8 == 0), "scale_cols_bf16: alignment");
    const long long n4 = rows * (long long)(cols / 4);
    long long grid = (n4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(scale_cols_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float4 *)w,
                       (const float4 *)scale, (uint2 *)out, n4, cols / 4);
    DETR_LAUNCH_CHECK("scale_cols_bf16");
    return 0;
}

extern "C" int detr_hip_colsum_f32(const float *x, float *out, int64_t rows, int32_t cols, int64_t ld, float alpha,
                                   void *stream) {
    DETR_REQUIRE(x && out && rows > 0 && cols > 0, "colsum: bad args");
    int rpb = 64;
    while ((rows + rpb - 1) / rpb > 2048) rpb *= 2;
    dim3 grid((unsigned)cdiv(cols, 256), (unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, (long long)rows, cols,
                       (long long)ld, alpha, rpb);
    DETR_LAUNCH_CHECK("colsum");
    return 0;
}

// deterministic column sums: partial[chunk][c] in a fixed per-thread order, then out[c] += alpha * (partial[0][c] + partial[1][c] + ...)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ x, float *__restrict__ part, long long rows, int cols,
                                                             long long ld, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
    float s0 = 0.f, s1 = 0.f;
    long long r = r0;
    for (; r + 1 < r1; r += 2) {
        s0 += x[r * ld + c];
        s1 += x[(r + 1) * ld + c];
    }
    if (r < r1) s0 += x[r * ld + c];
    part[(long long)blockIdx.y * cols + c] = s0 + s1;
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float *__restrict__ part, float *__restrict__ out, int chunks, int cols, float alpha) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += part[(long long)k * cols + c];
    out[c] += alpha * s;
}
static int colsum_det_rpb(int64_t rows) {
    int rpb = 64;
    while ((rows + rpb - 1) / rpb > 256) rpb *= 2;          // <= 256 chunks: the ordered finish pass stays short
    return rpb;
}
extern "C" int64_t detr_hip_colsum_det_scratch_floats(int64_t rows, int32_t cols) {
    if (rows <= 0 || cols <= 0) return -1;
    const int rpb = colsum_det_rpb(rows);
    return ((rows + rpb - 1) / rpb) * (int64_t)cols;
}
extern "C" int detr_hip_colsum_det_f32(const float *x, float *out, int64_t rows, int32_t cols, int64_t ld, float alpha, float *scratch,
                                       int64_t scratch_floats, void *stream) {
    DETR_REQUIRE(x && out && scratch && rows > 0 && cols > 0, "colsum_det: bad args");
    const int rpb = colsum_det_rpb(rows);
    const int chunks = (int)((rows + rpb - 1) / rpb);
    DETR_REQUIRE(scratch_floats >= (int64_t)chunks * cols, "colsum_det: scratch holds %lld floats, %lld needed", (long long)scratch_floats,
                 (long long)chunks * cols);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)cdiv(cols, 256), (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, x, scratch,
                       (long long)rows, cols, (long long)ld, rpb);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)cdiv(cols, 256)), dim3(256), 0, (hipStream_t)stream, scratch, out, chunks, cols, alpha);
    DETR_LAUNCH_CHECK("colsum_det");
    return 0;
}

extern "C" int detr_hip_add_bcast_f32(const float *x, const float *p, float *out, int64_t n, int64_t period,
                                      void *stream) {
    DETR_REQUIRE(x && p && out && n > 0 && period > 0, "add_bcast: bad args");
    DETR_REQUIRE(n % 4 == 0 && period % 4 == 0 && aligned16(x) && aligned16(p) && aligned16(out), "add_bcast: alignment");
    hipLaunchKernelGGL(add_bcast_kernel, dim3(ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)x,
                       (const float4 *)p, (float4 *)out, (long long)(n / 4), (long long)(period / 4));
    DETR_LAUNCH_CHECK("add_bcast");
    return 0;
}

extern "C" int detr_hip_add_f32(const float *a, const float *b, float *out, int64_t n, void *stream) {
    DETR_REQUIRE(a && b && out && n > 0, "add: bad args");
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)n);
    DETR_LAUNCH_CHECK("add");
    return 0;
}

extern "C" int detr_hip_sigmoid_bwd_f32(const float *dy, const float *y, float *dz, int64_t n, void *stream) {
    DETR_REQUIRE(dy && y && dz && n > 0, "sigmoid_bwd: bad args");
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dz,
                       (long long)n);
    DETR_LAUNCH_CHECK("sigmoid_bwd");
    return 0;
}

extern "C" int detr_hip_dropout_f32(const float *x, float *out, int64_t n, float p, uint32_t site, const uint32_t *step,
                                    void *stream) {
    DETR_REQUIRE(x && out && n > 0 && p >= 0.0f && p < 1.0f, "dropout: bad args");
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n,
                       1.0f / (1.0f - p), drop_thresh16(p), site, step);
    DETR_LAUNCH_CHECK("dropout");
    return 0;
}

namespace detr {
// n independent flat copies / accumulations described by a DEVICE table, one launch (blockIdx.y = entry): gathers the
// per-layer decoder cross-attention K / V weights into one [12*256, 256] operand and scatters its gradient back.
__global__ __launch_bounds__(256) void multi_copy_kernel(const detr_copy_entry *__restrict__ table) {
    const detr_copy_entry e = table[blockIdx.y];
    const uint4 *src = reinterpret_cast<const uint4 *>(e.src);
    uint4 *dst = reinterpret_cast<uint4 *>(e.dst);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < e.n16; i += (long long)gridDim.x * blockDim.x) {
        if (e.mode == 0) {
            dst[i] = src[i];
        } else {
            const float4 a = reinterpret_cast<const float4 *>(src)[i];
            float4 b = reinterpret_cast<float4 *>(dst)[i];
            b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
            reinterpret_cast<float4 *>(dst)[i] = b;
        }
    }
}
__global__ void set_u32x8_kernel(uint32_t *dst, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4, uint32_t v5,
                                 uint32_t v6, uint32_t v7) {
    if (threadIdx.x == 0) { dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3; dst[4] = v4; dst[5] = v5; dst[6] = v6; dst[7] = v7; }
}
}  // namespace detr

namespace detr {
// out[c] += scale[c] * sum_r x[r*ld + c] for an fp32 or bf16 matrix: the bias gradient of a conv whose frozen BN is folded
// into it (tf_backbone=True: keras.applications convs carry a trainable bias; y = scale*(conv + b) + ... -> db = scale * sum dz)
template <bool X16>
__global__ __launch_bounds__(256) void colsum_scaled_kernel(const void *__restrict__ xv, float *__restrict__ out, long long rows,
                                                            int cols, long long ld, const float *__restrict__ scale,
                                                            int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
    float s = 0.f;
    for (long long r = r0; r < r1; ++r) {
        if (X16) s += __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short *>(xv)[r * ld + c] << 16);
        else s += reinterpret_cast<const float *>(xv)[r * ld + c];
    }
    unsafeAtomicAdd(out + c, (scale ? scale[c] : 1.0f) * s);
}

// table-driven out[i] = a[i] * b[i] + c[i] for n small vectors in ONE launch (the effective BN shift of every conv with a bias)
__global__ __launch_bounds__(256) void fma_vec_group_kernel(const detr_fma_entry *__restrict__ table) {
    const detr_fma_entry e = table[blockIdx.x];
    for (int i = threadIdx.x; i < e.n; i += blockDim.x) e.out[i] = e.a[i] * e.b[i] + e.c[i];
}
}  // namespace detr

extern "C" int detr_hip_colsum_scaled(const void *x, int32_t x_dtype, int64_t rows, int32_t cols, int64_t ld, const float *scale,
                                      float *out, void *stream) {
    DETR_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols && (x_dtype == 0 || x_dtype == 1), "colsum_scaled: bad args");
    const int rpb = (int)((rows + 255) / 256 > 64 ? (rows + 255) / 256 : 64);
    dim3 grid((unsigned)cdiv(cols, 256), (unsigned)cdiv(rows, rpb));
    if (x_dtype == 1) hipLaunchKernelGGL(colsum_scaled_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, out, (long long)rows, cols, (long long)ld, scale, rpb);
    else hipLaunchKernelGGL(colsum_scaled_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, out, (long long)rows, cols, (long long)ld, scale, rpb);
    DETR_LAUNCH_CHECK("colsum_scaled");
    return 0;
}

extern "C" int detr_hip_fma_vec_group(const detr_fma_entry *table, int32_t n, void *stream) {
    DETR_REQUIRE(table && n > 0, "fma_vec_group: bad args");
    hipLaunchKernelGGL(fma_vec_group_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, table);
    DETR_LAUNCH_CHECK("fma_vec_group");
    return 0;
}

extern "C" int detr_hip_multi_copy(const detr_copy_entry *table, int32_t n, int32_t blocks_per_entry, void *stream) {
    DETR_REQUIRE(table && n > 0 && n <= 65535 && blocks_per_entry > 0, "multi_copy: bad args");
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)blocks_per_entry, (unsigned)n), dim3(256), 0, (hipStream_t)stream, table);
    DETR_LAUNCH_CHECK("multi_copy");
    return 0;
}

extern "C" int detr_hip_set_u32x8(uint32_t *dst, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4, uint32_t v5,
                                  uint32_t v6, uint32_t v7, void *stream) {
    DETR_REQUIRE(dst, "set_u32x8: null destination");
    hipLaunchKernelGGL(set_u32x8_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, v0, v1, v2, v3, v4, v5, v6, v7);
    DETR_LAUNCH_CHECK("set_u32x8");
    return 0;
}

extern "C" int detr_hip_relu_mask_f32(const float *g, const float *ref, float *out, int64_t n, void *stream) {
    DETR_REQUIRE(g && ref && out && n > 0, "relu_mask: bad args");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, g, ref, out,
                       (long long)n);
    DETR_LAUNCH_CHECK("relu_mask");
    return 0;
}

extern "C" int detr_hip_scale_cols_f32(const float *w, const float *scale, float *w_out, int64_t rows, int32_t cols,
                                       void *stream) {
    DETR_REQUIRE(w && scale && w_out && rows > 0 && cols > 0, "scale_cols: bad args");
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(scale_cols_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, w, scale, w_out, n,
                       cols);
    DETR_LAUNCH_CHECK("scale_cols");
    return 0;
}

extern "C" int detr_hip_scale_cols_t_f32(const float *w, const float *scale, float *w_out_t, int32_t rows, int32_t cols,
                                         void *stream) {
    DETR_REQUIRE(w && w_out_t && rows > 0 && cols > 0, "scale_cols_t: bad args");
    dim3 grid((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32));
    hipLaunchKernelGGL(scale_cols_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, scale, w_out_t, rows, cols);
    DETR_LAUNCH_CHECK("scale_cols_t");
    return 0;
}

extern "C" int detr_hip_bn_fold_f32(const float *weight, const float *bias, const float *mean, const float *var,
                                    float *scale, float *shift, int32_t C, float eps, void *stream) {
    DETR_REQUIRE(weight && bias && mean && var && scale && shift && C > 0, "bn_fold: bad args");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, weight, bias, mean, var,
                       scale, shift, C, eps);
    DETR_LAUNCH_CHECK("bn_fold");
    return 0;
}

extern "C" int detr_hip_axpy_f32(float *acc, const float *g, float a, int64_t n, void *stream) {
    DETR_REQUIRE(acc && g && n > 0, "axpy: bad args");
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, acc, g, a, (long long)n);
    DETR_LAUNCH_CHECK("axpy");
    return 0;
}
