/* detr_hip.h -- C ABI of libdetr_hip.so: the MI355X (gfx950) hot path of DETR
 * (ResNet backbone convolutions, transformer GEMMs/attention pieces, Hungarian set loss,
 * clip+Adam) as hand-written HIP kernels.
 *
 * The reference (Visual-Behavior/detr-tensorflow) has NO native/FFI boundary: its hot path is
 * TensorFlow ops plus one host callback into SciPy.  Each entry point below therefore cites the
 * reference Python lines whose arithmetic it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch.Tensor storage); nothing is
 *     allocated, freed or synchronised inside; every call only enqueues work on `stream`
 *     (a hipStream_t passed as void*) and is safe to capture in a hipGraph; whatever changes from one
 *     training step to the next (dropout step seed, Adam step sizes) is read from DEVICE memory, so a
 *     captured step can be replayed;
 *   - all arithmetic is fp32 ("f32" in the names); indices are int32 on the device;
 *   - return value: 0 = ok, negative = rejected (bad shape / alignment / launch failure); the
 *     reason is retrievable with detr_hip_last_error() (thread-local);
 *   - thread-safe for distinct streams.
 */
#ifndef DETR_HIP_H
#define DETR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DETR_HIP_ABI_VERSION 9

const char *detr_hip_last_error(void);
int detr_hip_abi_version(void);
/* The integer tuning variables DETR_HIP_<NAME> (A/B switches of the kernel dispatch; all default 0 = the measured heuristic) are
 * read from the environment ONCE, when the library is loaded -- never on the launch path.  A process that changes one afterwards
 * (tests, tuning scripts) calls this to have it re-read. */
int detr_hip_reload_tuning(void);
/* ABI self-check for hand-written mirrors of the descriptor structs below (ctypes / cgo / JNI): for struct `which`
 * (0 detr_reduce_desc, 1 detr_gemm_desc, 2 detr_conv3x3_desc, 3 detr_stem_desc, 4 detr_layernorm_desc, 5 detr_attn_desc,
 * 6 detr_setloss_desc, 7 detr_input_desc, 8 detr_postprocess_desc) writes out[0] = sizeof, out[1..] = offsetof of every field in
 * declaration order, and returns the number of values (negative: unknown struct). */
int detr_hip_struct_layout(int32_t which, int32_t *out, int32_t cap);
/* zero fill of accumulators (an ordinary kernel: a captured hipMemsetAsync node did not replay reliably) */
int detr_hip_memset_zero(void *ptr, size_t bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Generic batched GEMM with fused epilogue:   C = epi( A(MxK) * B(KxN) )
 *   replaces: every tf.matmul / 1x1 Conv2D / Linear on the path --
 *     detr_tf/networks/custom_layers.py:49-50 (Linear), resnet_backbone.py:101,107,111 (1x1 convs),
 *     detr.py:44 (input_proj), transformer.py:294-304,317,343,346 (MHA projections, QK^T, PV),
 *     and their tape gradients (optimizers.py:115): dgrad and wgrad are the same kernel with
 *     other operand layouts.
 *   epilogue, in this order:  v = acc; v *= scale[n]; v += bias[n]; v *= alpha;
 *                             v += residual[m,n]; act(v); v = mask[m,n] > 0 ? v : 0
 *     (frozen-BN fold custom_layers.py:21-24, bias, q-scaling transformer.py:307, residual adds,
 *      ReLU, sigmoid detr.py:188, ReLU-backward masking)
 *   a_kcontig = 1: A[m*lda + k]   0: A[k*lda + m]        (same for B with n)
 *   batch z in [0,batch): z0 = z / batch_inner, z1 = z % batch_inner, operand offset z0*s?0 + z1*s?1
 *   split_k > 1: the K range is split over split_k workgroups whose partial results are
 *     atomically ADDED to C (C must hold zeros or the value to accumulate onto); only
 *     scale and alpha are allowed in the epilogue then.
 * ------------------------------------------------------------------------------------------- */
/* one deferred deterministic split-K reduction: C[r][c] += alpha * scale[c] * sum_s ws[s*part_stride + r*cols + c]
 * (+ the fused bias gradient rs_out[r] += rs_alpha * sum_s rs_ws[s*rows + r]); filled by detr_hip_gemm_f32 /
 * detr_hip_gemm_group_f32 when detr_gemm_desc.defer_out is set, consumed by detr_hip_splitk_reduce_many */
typedef struct detr_reduce_desc {
    const float *ws; int32_t splits; int64_t part_stride; int32_t rows, cols;
    float *C; int64_t ldc; float alpha; const float *scale;
    const float *rs_ws; float *rs_out; float rs_alpha;
    /* ABI 5: slab layout.  ts_bm == 0: row-major slabs (ws[s*part_stride + r*cols + c]).  ts_bm > 0: tile-ordered slabs as the
     * split kernels of this library write them when the workspace holds detr_hip_workspace_bytes_* bytes -- tiles of
     * ts_bm x ts_bn (ts_tiles_n of them per tile row), each a lane-linear image of the MFMA accumulator registers
     * (csrc/gemm_core.h: store_slab_ts); part_stride then counts the padded tile grid. */
    int32_t ts_bm, ts_bn, ts_tiles_n;
} detr_reduce_desc;

typedef struct {
    int32_t M, N, K;
    const float *A; int64_t lda; int32_t a_kcontig;
    const float *B; int64_t ldb; int32_t b_kcontig;
    float *C; int64_t ldc;
    int32_t batch, batch_inner;
    int64_t sA0, sA1, sB0, sB1, sC0, sC1;
    float alpha;
    const float *scale;
    const float *bias;
    const float *residual; int64_t ldr;
    const float *mask; int64_t ldmask;
    int32_t act;            /* 0 none, 1 relu, 2 sigmoid */
    int32_t split_k;        /* 0/1 = no split */
    /* optional caller-provided scratch for DETERMINISTIC split-K: when it holds at least
     * split_k*M*N floats (and batch == 1) every split stores its partial tile there with plain
     * 16-byte stores and a second launch reduces them onto C (C += alpha*scale*sum); otherwise
     * the partials are accumulated with fp32 atomics. */
    float *workspace; int64_t workspace_bytes;
    /* fused dropout (training mode, transformer.py:169,174-176): keep-mask = keyed counter hash of
     * (site = dropout_seed, step seed = *dropout_step, element row*N + col), kept values scaled by 1/(1-p); applied
     * before the residual add when a residual is given, otherwise after the activation.  0 = off.
     * dropout_step (last field of this struct) is a DEVICE pointer to the uint32 seed of the current training step
     * (NULL = 0): a hipGraph replay of a captured step reads a fresh seed. */
    float dropout_p; uint32_t dropout_seed;
    /* 0 = exact fp32 MFMA (parity mode); 1 = bf16 MFMA with fp32 storage / accumulation (BASELINE config C3:
     * operands are rounded to bf16 on their way into LDS); 2 (ABI 9, "f32x3") = fp32 storage AND fp32 accuracy on the bf16
     * matrix pipe: every fp32 operand value is split exactly into three bf16 values (x = h + m + l) and a product is the sum
     * of the six largest of the nine bf16 partial products, accumulated in fp32 (relative error of a product <= 2^-23, the
     * size of one fp32 rounding; csrc/gemm_core.h: mma_ktile_split3) -- 6 bf16 MFMAs instead of 8 fp32 MFMAs per 32x32x16
     * block at 16x the instruction rate.  Operands, C, residual and mask must be fp32 (the *_dtype fields 0); tiles other
     * than 64x64 / 128x128 (N <= 32) run the exact kernel. */
    int32_t compute;
    /* optional fused bias gradient of a weight-gradient GEMM (dW = dy^T x): rowsum_a[m] += rowsum_alpha * sum_k A[m][k]
     * (A must be MN-contiguous, a_kcontig = 0; batch == 1; with split_k > 1 the deterministic workspace path is
     * required and the workspace must hold split_k*(M*N + M) floats).  Replaces the reference's separate
     * reduce_sum for every Linear bias gradient (tape gradient of custom_layers.py Linear bias). */
    float *rowsum_a; float rowsum_alpha;
    /* 0: B is fp32.  1: B points to bf16 data (uint16, RNE-rounded weights; ldb in elements) -- bf16 compute only: the
     * engine keeps a per-step bf16 shadow of the weights so that the weight operand is read at half the bytes and
     * without conversion (K %% 8 == 0 for a k-contiguous B, N %% 4 == 0 otherwise). */
    int32_t b_dtype;
    /* bf16 STORAGE of activation tensors (bf16 compute only; the engine's backbone tensors when DETR_HIP_ACT16=1): 1 = the
     * tensor is bf16 in memory (uint16, RNE; leading dimensions in elements), 0 = fp32.  The epilogue arithmetic stays
     * fp32 and the result is rounded once.  Not with split_k / batch (partial slabs and gradients of parameters stay fp32). */
    int32_t a_dtype, c_dtype, r_dtype, m_dtype;     /* A, C, residual, mask (m_dtype 2: bit-packed, see maskbits_out) */
    const uint32_t *dropout_step;                   /* see dropout_p */
    /* optional (HOST pointer): with deterministic split-K, do not launch the reduction but describe it here (splits = 0 when
     * no reduction is pending); the caller keeps `workspace` intact and reduces many slabs per launch with
     * detr_hip_splitk_reduce_many -- the weight gradients of a backward pass are only needed by the optimiser */
    detr_reduce_desc *defer_out;
    /* ABI 5: bit-packed ReLU masks (bf16 activation storage only, N %% 8 == 0).  The backward of `relu` only needs the SIGN of the
     * saved activation: a GEMM whose epilogue ends in the ReLU can emit, next to C, one byte per 8 outputs --
     * bit (n & 7) of maskbits_out[m * ld_maskbits_out + n / 8] = (C[m][n] > 0) as stored -- and a later GEMM takes it instead of
     * the activation tensor: `mask` pointing to those bytes with m_dtype = 2 and ldmask = their row pitch in BYTES (1/16 of the
     * bytes of the bf16 tensor on an HBM-bound launch; resnet_backbone.py:132-135 backward). */
    uint8_t *maskbits_out; int64_t ld_maskbits_out;
    /* Optional fused LayerNormalization of the output rows (ln_y != NULL; transformer.py:151-152 behind every attention / FFN block,
     * :169-170,177,215-233): C = drop((A B^T + bias) * alpha) + residual is written as usual (the backward reads the LayerNorm INPUT)
     * and, from the same launch,  ln_y = LayerNorm(C; ln_gamma, ln_beta, ln_eps), ln_mean / ln_rstd [M], optionally
     * ln_y2[r] = ln_y[r] + ln_add[r % ln_add_rows] and the bf16 twin ln_y16 -- the outputs of detr_hip_layernorm_fwd, bit for bit.
     * Needs: compute = 1, N = 256 = ldc (contiguous fp32 C / ln_y / ln_y2 / residual rows), a_kcontig = b_kcontig = 1, b_dtype = 1,
     * K % 8 == 0, batch = split_k = 1, act = 0, no scale / mask; anything else is rejected. */
    const float *ln_gamma, *ln_beta; float *ln_y, *ln_mean, *ln_rstd; const float *ln_add; int32_t ln_add_rows; float *ln_y2;
    uint16_t *ln_y16; float ln_eps;
} detr_gemm_desc;
int detr_hip_gemm_f32(const detr_gemm_desc *d, void *stream);
/* the kernel family detr_hip_gemm_f32 would launch for this descriptor (no launch): 0 tile engine, 1 streaming kernel, 2 ring kernel (bf16),
 * 3 ring kernel (fp32), 4 GEMM + LayerNorm, 5 ring weight gradient; negative = rejected.  For callers that bill launches to roofline families. */
int detr_hip_gemm_family(const detr_gemm_desc *d);
int detr_hip_splitk_reduce_many(const detr_reduce_desc *descs, int32_t n, void *stream);
/* n independent GEMMs in one call.  Consecutive members (up to 4) that share one kernel variant -- 64x64 tiles, same operand
 * layouts / storage types, no batch -- are issued as ONE launch (members on blockIdx.y) and their split-K reductions as one
 * more; otherwise the members are launched one after the other.  Members must not alias each other's outputs or workspaces.
 * Used for the Q / K / V projections of an attention block and their gradients (transformer.py:297-309), whose individual
 * launches fill only 1-2 workgroups per CU. */
int detr_hip_gemm_group_f32(const detr_gemm_desc *descs, int32_t n, void *stream);

/* ---------------------------------------------------------------------------------------------
 * 3x3 convolution (explicit zero pad `pad`, then VALID, stride 1|2, dilation 1), NHWC / HWIO,
 * as an implicit GEMM (no im2col buffer):
 *   replaces: ZeroPadding2D + Conv2D(3x3) + FrozenBatchNorm2D + ReLU
 *             detr_tf/networks/resnet_backbone.py:98,104-105,123-126 and its gradients.
 *   mode 0 fwd  : y[N,Ho,Wo,Co]  = epi( conv(x[N,Hi,Wi,Ci], w[3,3,Ci,Co]) )
 *   mode 1 dgrad: dx[N,Hi,Wi,Ci] = epi( conv^T(dy[N,Ho,Wo,Co], w) )          (x = dy in, y = dx out)
 *   mode 2 wgrad: dw[3,3,Ci,Co] += scale[co] * sum_m x[pix(m,tap),ci] * dy[m,co]   (atomic; y = dw)
 *   epilogue fields as in detr_gemm_desc (rows = output pixels, ld = channel count).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N, Hi, Wi, Ci, Ho, Wo, Co, stride, pad;
    const float *x;      /* fwd: input; dgrad: dy; wgrad: input x */
    const float *w;      /* fwd/dgrad: weights; wgrad: dy */
    float *y;            /* fwd: output; dgrad: dx; wgrad: dw (pre-zeroed / accumulated onto) */
    float alpha;
    const float *scale;
    const float *bias;
    const float *residual;
    const float *mask;
    int32_t act;
    int32_t split;       /* wgrad: number of row splits (0 = auto) */
    float *workspace; int64_t workspace_bytes;   /* wgrad: scratch for deterministic split reduction (see detr_gemm_desc) */
    int32_t compute;     /* 0 = exact fp32 MFMA, 1 = bf16 MFMA, 2 = fp32 accuracy on the bf16 matrix pipe (see detr_gemm_desc) */
    int32_t w_dtype;     /* modes 0/1 with compute = 1: 1 = `w` points to bf16 data (weight shadow, see detr_gemm_desc.b_dtype) */
    /* bf16 storage of the activation tensors (see detr_gemm_desc.a_dtype): x = the tensor passed as `x`, y = the one passed as `y`
     * (mode 2: x = input activations, `w` = dy whose dtype is w_dtype, y = dw always fp32), r = residual, m = mask */
    int32_t x_dtype, y_dtype, r_dtype, m_dtype;
    /* ABI 5: bit-packed ReLU masks as in detr_gemm_desc (bf16 tensors only): mode 0 with act = 1 also writes one byte per 8 output
     * channels, bit = (y > 0), row pitch Co / 8 bytes per pixel; modes 0 / 1 take `mask` as such bytes when m_dtype = 2 */
    uint8_t *maskbits_out;
} detr_conv3x3_desc;
int detr_hip_conv3x3_f32(const detr_conv3x3_desc *d, int32_t mode, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Stem helpers (detr_tf/networks/resnet_backbone.py:11-26):
 *   3x3 stride-2 max pool over the ZERO-padded (pad 1) map (padding takes part in the max, as
 *   ZeroPadding2D + MaxPool2D('valid') does) with argmax for the backward;
 *   maxpool backward fused with the stem ReLU mask.
 * ------------------------------------------------------------------------------------------- */
/* fp32 -> bf16 (RNE) copies that feed the bf16 weight operands: a flat conversion (n %% 4 == 0) and the frozen-BN fold
 * out[r][c] = bf16(w[r][c] * scale[c]) (custom_layers.py:21-24 folded into the conv kernel). */
int detr_hip_cvt_bf16(const float *x, uint16_t *out, int64_t n, void *stream);
int detr_hip_scale_cols_bf16(const float *w, const float *scale, uint16_t *out, int64_t rows, int32_t cols, void *stream);
/* The same fold for a whole list of kernels in ONE launch (the backbone has 52 folded convs and refolds them after every
 * optimiser step).  `table` is DEVICE memory holding n entries of this layout; c4 = cols/4, n4 = rows*cols/4. */
typedef struct detr_scale_entry {
    const float *w;
    const float *scale;
    uint16_t *out;
    int64_t n4;
    int32_t c4;
    int32_t reserved;
} detr_scale_entry;
int detr_hip_scale_cols_bf16_group(const detr_scale_entry *table, int32_t n, void *stream);

/* The stem convolution as an IMPLICIT GEMM (no im2col buffer): ZeroPadding2D(3) + 7x7 stride-2 VALID conv 3 -> 64
 * (+ folded frozen BN + ReLU), resnet_backbone.py:11-26.   mode 0: y[M,64] = act((gather(img) @ w[147][64]) * scale + bias) * ...
 * with the GEMM epilogue order of detr_gemm_desc; mode 2 (weight gradient): w = dy [M,64], y = dw [147][64],
 * dw += alpha * scale[co] * sum_m gather(img)[m][k] * dy[m][co], reduction split over `split` workgroups through
 * `workspace` (split*147*64 floats, deterministic).  M = N*Ho*Wo, k = (kh*7 + kw)*3 + c. */
typedef struct {
    int32_t N, H, W, Ho, Wo;
    const float *img;       /* [N, H, W, 3] */
    const float *w;         /* mode 0: kernel [147][64]; mode 2: dy [M][64] */
    float *y;               /* mode 0: output [M][64]; mode 2: dw [147][64] */
    float alpha;
    const float *scale;
    const float *bias;
    int32_t act;
    int32_t split;
    float *workspace; int64_t workspace_bytes;
    int32_t compute;        /* 0 = exact fp32 MFMA, 1 = bf16 MFMA, 2 = fp32 accuracy on the bf16 matrix pipe (see detr_gemm_desc) */
    int32_t w_dtype;        /* mode 2: 1 = dy is bf16 in memory (bf16 activation storage) */
    int32_t y_dtype;        /* mode 0: 1 = the output is stored as bf16 */
} detr_stem_desc;
int detr_hip_stem_conv7x7_f32(const detr_stem_desc *d, int32_t mode, void *stream);
int detr_hip_maxpool3x3s2_fwd_f32(const float *x, float *y, uint8_t *argmax, int32_t N, int32_t H,
                                  int32_t W, int32_t C, int32_t Ho, int32_t Wo, void *stream);
/* dx[n,h,w,c] = (x[n,h,w,c] > 0) * sum over windows whose argmax is (h,w) of dy */
int detr_hip_maxpool3x3s2_bwd_f32(const float *dy, const uint8_t *argmax, const float *x, float *dx,
                                  int32_t N, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                                  void *stream);
/* stride-2 1x1 downsample helpers (resnet_backbone.py:111-113): gather / zero-filled scatter */
/* bf16-storage twins of the stem max pooling (see detr_gemm_desc.a_dtype); the subsample kernels are plain 16-byte copies and
 * serve bf16 tensors with C/2 "float" channels. */
int detr_hip_maxpool3x3s2_fwd_bf16(const uint16_t *x, uint16_t *y, uint8_t *argmax, int32_t N, int32_t H, int32_t W,
                                   int32_t C, int32_t Ho, int32_t Wo, void *stream);
int detr_hip_maxpool3x3s2_bwd_bf16(const uint16_t *dy, const uint8_t *argmax, const uint16_t *x, uint16_t *dx, int32_t N,
                                   int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo, void *stream);
int detr_hip_subsample2_fwd_f32(const float *x, float *y, int32_t N, int32_t H, int32_t W, int32_t C,
                                int32_t Ho, int32_t Wo, void *stream);
int detr_hip_subsample2_bwd_f32(const float *dy, float *dx, int32_t N, int32_t H, int32_t W, int32_t C,
                                int32_t Ho, int32_t Wo, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Row kernels of the transformer (detr_tf/networks/transformer.py):
 *   LayerNormalization(eps) over the last dim C (C % 4 == 0, C <= 1024) :151-152,169-170,177
 *   softmax over rows of the score tensor :340, and their backwards;
 *   column sums (bias gradients), broadcast add (src + pos :161-163, tgt + query_pos :209),
 *   sigmoid backward (detr.py:188).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t rows, C; float eps;
    const float *x;                 /* fwd: input rows; bwd: the forward INPUT (xhat is recomputed from mean / rstd) */
    const float *gamma, *beta;
    float *y;                       /* fwd output */
    float *mean, *rstd;             /* [rows]: written by fwd, read by bwd */
    /* fwd, optional fused second output  y2[r] = y[r] + add[r % add_rows]  -- the `src + pos` / `tgt + query_pos`
     * operand of the NEXT attention block (transformer.py:161-163,209,219), written while y is still in registers */
    const float *add; int32_t add_rows; float *y2;
    uint16_t *y16;                  /* fwd, optional: bf16 (RNE) twin of y -- the A operand of the next bf16-compute GEMM */
    /* bwd */
    const float *dy; float *dx; float *dgamma, *dbeta;     /* dgamma / dbeta are ACCUMULATED */
    /* dgamma/dbeta reduction: deterministic (per-block partials in `workspace` + a finish launch) when the workspace
     * holds at least 512*2*C floats, fp32 atomics otherwise (workspace may be NULL) */
    float *workspace; int64_t workspace_bytes;
    const float *dx_add;            /* optional: dx += dx_add (a second gradient branch into the same tensor) */
    /* optional fused second output dx_drop[i] = keep(site, step, i) ? dx[i] / (1-p) : 0 -- the gradient through the
     * Dropout that precedes the residual add in front of this LayerNorm (transformer.py:169,176,215,226,232), i.e. the
     * mask of the forward GEMM epilogue regenerated on the gradient (element index row*C + col) */
    float *dx_drop; float dropout_p; uint32_t dropout_site; const uint32_t *dropout_step;
    uint16_t *dx_drop16;            /* optional bf16 twin of dx_drop (dropout_p = 0: of dx) */
    /* optional (HOST pointer; needs the workspace path): do not launch the dgamma / dbeta finish but report the number of
     * partial blocks -- the caller owns `workspace` ([blocks][2*C]: gamma partials, then beta partials, per block) and
     * reduces it later through detr_hip_splitk_reduce_many (two entries: rows 1, cols C, splits = blocks, part_stride 2*C) */
    int32_t *defer_blocks_out;
    const float *dy_add;            /* bwd, optional: the incoming gradient is dy + dy_add (two branches meeting at this LayerNorm's output) */
} detr_layernorm_desc;
int detr_hip_layernorm_fwd(const detr_layernorm_desc *d, void *stream);
int detr_hip_layernorm_bwd(const detr_layernorm_desc *d, void *stream);
/* out[c] += alpha * sum_r x[r*ld + c] (atomic) */
int detr_hip_colsum_f32(const float *x, float *out, int64_t rows, int32_t cols, int64_t ld, float alpha,
                        void *stream);
/* The same sum in a FIXED order (round 5: the bias gradients of input_proj and of the heads -- detr.py:172-175, 192-203 -- were the one
 * gradient of the step that differed between two runs, by float-atomic order): per-chunk partial sums into `scratch` (at least
 * detr_hip_colsum_det_scratch_floats(rows, cols) floats), then one ordered pass over the chunks.  Two launches, no atomics. */
int64_t detr_hip_colsum_det_scratch_floats(int64_t rows, int32_t cols);
int detr_hip_colsum_det_f32(const float *x, float *out, int64_t rows, int32_t cols, int64_t ld, float alpha, float *scratch,
                            int64_t scratch_floats, void *stream);
/* Backward of a bottleneck 1x1 convolution 64 -> 256 channels in ONE pass over dY (round 5; reference: resnet_backbone.py:116-137 -- conv3 and the
 * projection shortcut of a BottleNeck -- and optimizers.py:110-120): da = (use_mask ? (a > 0) : 1) . (dY W^T) and dW += alpha * scale[n] * (a^T dY).
 * dY [M][ldg] (d2 = 256 columns), a [M][lda] (d1 = 64 columns: weight-gradient operand AND ReLU mask), W [64][ldw] (256 columns contiguous),
 * da [M][ldda] -- bf16; dW [64][lddw] fp32, scale [256] fp32 or NULL.  workspace: at least detr_hip_conv1x1_bwd_fused_workspace_floats(M) floats
 * (one 64 x 256 partial per workgroup, summed in a fixed order).  da is bit-identical to detr_hip_gemm_f32 on the same operands. */
int64_t detr_hip_conv1x1_bwd_fused_workspace_floats(int64_t M);
int detr_hip_conv1x1_bwd_fused_bf16(const uint16_t *dy, int64_t ldg, const uint16_t *a, int64_t lda, const uint16_t *w, int64_t ldw, uint16_t *da,
                                    int64_t ldda, int32_t use_mask, float *dw, int64_t lddw, const float *scale, float alpha, int64_t M, int32_t d1,
                                    int32_t d2, float *workspace, int64_t workspace_floats, void *stream);
/* out[i] = x[i] + p[i % period]   (n, period multiples of 4) */
int detr_hip_add_bcast_f32(const float *x, const float *p, float *out, int64_t n, int64_t period,
                           void *stream);
/* out[i] = a[i] + b[i] */
int detr_hip_add_f32(const float *a, const float *b, float *out, int64_t n, void *stream);
/* dz[i] = dy[i] * y[i] * (1 - y[i]) */
int detr_hip_sigmoid_bwd_f32(const float *dy, const float *y, float *dz, int64_t n, void *stream);
/* out[i] = keep(site, *step, i) ? x[i] / (1-p) : 0   -- the dropout mask of detr_gemm_desc regenerated on a gradient */
int detr_hip_dropout_f32(const float *x, float *out, int64_t n, float p, uint32_t site, const uint32_t *step, void *stream);
/* n independent flat copies (mode 0: 16-byte units, any element type) or fp32 accumulations dst += src (mode 1) in ONE launch;
 * `table` is DEVICE memory.  Gathers the per-layer decoder cross-attention K / V projection weights (transformer.py:221-223,
 * the memory operand is layer-invariant) into one [layers*256, 256] GEMM operand and scatters its gradient back. */
typedef struct detr_copy_entry {
    const void *src;
    void *dst;
    int64_t n16;        /* number of 16-byte units */
    int32_t mode;       /* 0 copy, 1 fp32 accumulate */
    int32_t reserved;
} detr_copy_entry;
int detr_hip_multi_copy(const detr_copy_entry *table, int32_t n, int32_t blocks_per_entry, void *stream);
/* tf_backbone=True (tf.keras.applications ResNet50, detr.py:146-148: convs WITH a trainable bias under a frozen BatchNorm):
 *   out[c] += scale[c] * sum_r x[r*ld + c]   (x fp32: x_dtype 0, bf16: 1) -- the bias gradient through the folded BN;
 *   out[i] = a[i]*b[i] + c[i] for n small vectors in one launch (table in DEVICE memory) -- the effective shift
 *   beta - mean*scale + scale*bias of every conv, rebuilt when the biases move. */
typedef struct detr_fma_entry {
    const float *a, *b, *c;
    float *out;
    int32_t n;
    int32_t reserved;
} detr_fma_entry;
int detr_hip_colsum_scaled(const void *x, int32_t x_dtype, int64_t rows, int32_t cols, int64_t ld, const float *scale,
                           float *out, void *stream);
int detr_hip_fma_vec_group(const detr_fma_entry *table, int32_t n, void *stream);
/* dst[0..7] = v0..v7 (uint32; the values travel as kernel arguments): the per-step dropout seed */
int detr_hip_set_u32x8(uint32_t *dst, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4, uint32_t v5,
                       uint32_t v6, uint32_t v7, void *stream);
/* out[i] = (ref[i] > 0) ? g[i] : 0 */
int detr_hip_relu_mask_f32(const float *g, const float *ref, float *out, int64_t n, void *stream);
/* w_out[k, co] = w[k, co] * scale[co]   (frozen-BN scale folded into conv kernels, HWIO flat) */
int detr_hip_scale_cols_f32(const float *w, const float *scale, float *w_out, int64_t rows, int32_t cols,
                            void *stream);
/* w_out_t[co, k] = w[k, co] * scale[co] (scale may be NULL): transposed copy, i.e. a K-contiguous GEMM B operand */
int detr_hip_scale_cols_t_f32(const float *w, const float *scale, float *w_out_t, int32_t rows, int32_t cols,
                              void *stream);
/* frozen BN vectors -> (scale, shift)  custom_layers.py:21-23 */
int detr_hip_bn_fold_f32(const float *weight, const float *bias, const float *mean, const float *var,
                         float *scale, float *shift, int32_t C, float eps, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused multi-head attention core, head_dim 32 (detr_tf/networks/transformer.py:308-345):
 *   o[b,t,h*32:+32] = softmax_s( q[b,t,h] . k[b,s,h] ) v[b,s,h]      (scale = head_dim**-0.5, :307)
 * on batch-first token matrices; the [T,S] probability tensor is never
 * written to memory.  lse [B*H, T] (log-sum-exp per score row) is saved for the backward, which
 * recomputes the probabilities; delta [B*H, T] is scratch (rowsum(dO*O)).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B, H, T, S;
    /* batch-first token matrices; every tensor has its OWN row stride (floats), so q / k / v (and their gradients) may
     * be column blocks of one packed projection buffer: [rows, 768] for self-attention, [rows, layers*256] for the
     * layer-invariant decoder cross-attention K / V of all layers */
    const float *q; int64_t ldq;
    const float *k; int64_t ldk;
    const float *v; int64_t ldv;
    float *o; int64_t ldo;          /* fwd: output; bwd: the forward output */
    float *lse;                     /* [B*H, T] */
    const float *d_o; int64_t ldd_o;
    float *dq; int64_t lddq;
    float *dk; int64_t lddk;
    float *dv; int64_t lddv;
    float *delta;                   /* [B*H, T] scratch */
    /* softmax(scale * q.k): the reference scales the projected query (WQ *= head_dim**-0.5, transformer.py:307); here
     * the factor is folded into the constant q is loaded with, and dq is the gradient w.r.t. the UNSCALED q */
    float scale;
    /* attention-probability dropout (transformer.py:341): element index ((b*H + h)*T + t)*Sp + s, Sp = S rounded up to
     * even (one keyed counter hash per pair of adjacent keys, csrc/common.h); site / step as in detr_gemm_desc */
    float dropout_p; uint32_t dropout_site; const uint32_t *dropout_step;
    /* 0 = exact fp32 MFMA; 1 = bf16 MFMA operands (Q, K, V, P, dO, dS rounded to bf16; accumulation, softmax
     * statistics, LSE, delta and outputs fp32; csrc/attention_bf16.hip) */
    int32_t compute;
    /* ABI 8.  io_dtype = 1 (compute = 1 only; csrc/attention_dma.hip): q, k, v, o, d_o, dq, dk, dv point to bf16 data (uint16, RNE;
     * row strides in elements, multiples of 8), `q` holds bf16(scale * log2(e) * q_unscaled) -- the projection GEMM's alpha -- and dq
     * is still the gradient w.r.t. the UNSCALED q; lse stays fp32 and `delta` must hold 2*B*H*T floats (delta / keep-scale and
     * lse * log2(e) - log2(keep-scale), handed from the dQ kernel to the dK / dV kernel).  With dropout_p > 0 the keep flags are read
     * as BITS from `dropmask` (detr_hip_attention_dropmask_words(B, H, T, S) uint32 words: one word per (query, 32-key tile), word
     * (bh * ceil(S/32) + tile) * 32 ceil(T/32) + query, bit = key), which detr_hip_attention_dropmask fills
     * from (dropout_site, *dropout_step) -- the same function as the keyed counter hash above, evaluated once per step instead of in
     * each of the three kernels. */
    int32_t io_dtype;
    uint32_t *dropmask;
} detr_attn_desc;
int detr_hip_attention_fwd(const detr_attn_desc *d, void *stream);
int detr_hip_attention_bwd(const detr_attn_desc *d, void *stream);
/* keep bits of the attention-probability dropout of one site and step (io_dtype = 1): uses B, H, T, S, dropout_p, dropout_site,
 * dropout_step and dropmask of the descriptor */
int64_t detr_hip_attention_dropmask_words(int32_t B, int32_t H, int32_t T, int32_t S);
int detr_hip_attention_dropmask(const detr_attn_desc *d, void *stream);
/* the same for n sites of one step in ONE launch (they must share B*H, dropout_p and dropout_step; a DETR step has 18 sites) */
int detr_hip_attention_dropmask_many(const detr_attn_desc *descs, int32_t n, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Hungarian set loss (detr_tf/loss/hungarian_matching.py:163-203, detr_tf/loss/loss.py:22-179,
 * detr_tf/bbox.py:29-124,171-183).  P = levels * B problems; prediction p of level lv, image b:
 *   logits + lv*sL_l + b*sL_b + q*sL_q + c ,  boxes + lv*sB_l + b*sB_b + q*sB_q + k
 * targets in the reference's header layout (detr_tf/data/processing.py:35-55):
 *   t_bbox [B, R, 4] f32 (row 0 = [n,0,0,0]), t_class [B, R] int64 (row 0 = 0).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t levels, B, Q, C, R;        /* R = target rows incl. header (100) */
    const float *logits; int64_t sL_l, sL_b, sL_q;
    const float *boxes;  int64_t sB_l, sB_b, sB_q;
    const float *t_bbox; const int64_t *t_class;
    int32_t background_class;
} detr_setloss_desc;

/* K12: cost[p][q][j] (ld = R-1 per q) = 5*L1 + 1*(-softmax[class_j]) + 2*(-GIoU)   hungarian_matching.py:171-195 */
int detr_hip_match_cost_f32(const detr_setloss_desc *d, float *cost, void *stream);
/* K13: exact rectangular assignment per problem (replaces the tf.numpy_function -> SciPy call,
 *   hungarian_matching.py:27-46,197).  cost [P][Q][ldc] f32, n[P] taken from the t_bbox headers
 *   (n_from_header[b*R*4]) ; outputs tgt_for_pred [P][Q] int32 (-1 = unmatched prediction = the
 *   complement of the reference's bool selector) and pred_for_tgt [P][ldc] int32 (-1 padded).
 *   status[P]: 0 ok, 1 infeasible/NaN (the reference raises there). */
int detr_hip_assign_f32(const float *cost, int32_t P, int32_t Q, int32_t ldc, const float *t_bbox,
                        int32_t B, int32_t R, int32_t *tgt_for_pred, int32_t *pred_for_tgt,
                        int32_t *status, void *stream);
/* K14: loss.py:37-96.  Three steps so that data-parallel ranks can all-reduce `sums` in between:
 *   sums [levels][8] = {sum w*CE, -, n_neg_correct, n_neg, n_pos, n_pos_not_bg, n_pos_correct, -}
 *                      {.. [1], [7] unused: the normaliser sum w is derived as 0.1*n_neg + n_pos from the counts, which
 *                      float atomics accumulate exactly in any order -- the gradients are then a deterministic function
 *                      of the matching};  box sums [levels][2] are stored at sums + levels*8:
 *                      {sum L1, sum (1-GIoU)} ; all ACCUMULATED atomically (zero them first). */
int detr_hip_set_loss_sums_f32(const detr_setloss_desc *d, const int32_t *tgt_for_pred, float *sums,
                               void *stream);
/* losses [levels][6] = label_cost,true_neg,true_pos,pos_accuracy,giou_loss,l1_loss (loss.py:172-179);
 * total[0] = sum_lv 1*label + 2*giou + 5*l1 (loss.py:6-19) */
int detr_hip_set_loss_finalize_f32(const float *sums, int32_t levels, float *losses, float *total,
                                   void *stream);
/* gradients of loss_scale*total w.r.t. logits and boxes (same strides as the inputs) */
int detr_hip_set_loss_grad_f32(const detr_setloss_desc *d, const int32_t *tgt_for_pred, const float *sums,
                               float loss_scale, float *d_logits, float *d_boxes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Device-side input stage (the step's producer side; detr_tf/data/processing.py:6-21,35-55, data/transformation.py:82-91).
 *   detr_hip_input_stage: uint8 NHWC batch -> fixed-size resize (cv2.resize semantics in fp32: half-pixel centres,
 *     replicated border, cubic a = -0.75; rounded + saturated back to uint8) -> normalisation through a [3][256] float
 *     lookup table the host builds in float64 exactly as `normalized_images` does (dst channel c = lut[c][src channel
 *     perm[c]]; "tf_resnet" = BGR order, perm {2,1,0}) -> float32 NHWC.
 *   detr_hip_pad_labels: ragged targets (boxes [N,4] cx,cy,w,h; classes [N]; offsets [B+1]) -> the reference's padded
 *     layout with the in-band header row: t_bbox [B,R,4] (row 0 = [n,0,0,0]), t_class [B,R] (row 0 = 0), zero padded.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B, Hs, Ws, Hd, Wd;
    const uint8_t *src; int64_t src_batch_stride;    /* bytes between images */
    float *dst;                                      /* [B, Hd, Wd, 3] */
    const float *lut;                                /* [3][256] */
    int32_t perm[3];
    int32_t interpolation;                           /* 0 nearest, 1 linear, 2 cubic (imgaug's Resize default) */
} detr_input_desc;
int detr_hip_input_stage(const detr_input_desc *d, void *stream);
int detr_hip_pad_labels(const float *boxes, const int64_t *classes, const int32_t *offsets, int32_t B, int32_t R,
                        float *t_bbox, int64_t *t_class, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Batched post-processing (detr_tf/inference.py:68-95, which handles batch element 0 only; the validation loops call it
 * once per image, logger/training_logging.py:61-88, eval.py:41-55).  One launch for the whole batch: per query
 * score = max softmax probability, label = first arg-max of the probabilities; queries whose label is `background_class`
 * are dropped (order preserved); boxes converted (bbox.py:171-196: xyxy / yxyx are clipped to [0, 1]); the kept detections
 * of image b are written compacted to out_*[b][0 .. counts[b]).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B, Q, C;
    const float *logits; int64_t sL_b, sL_q;     /* [B, Q, C], unit class stride */
    const float *boxes;  int64_t sB_b, sB_q;     /* [B, Q, 4] cx, cy, w, h */
    int32_t background_class;
    int32_t bbox_format;                         /* 0 "xy_center", 1 "xyxy", 2 "yxyx" */
    float *out_boxes;                            /* [B, Q, 4] */
    int64_t *out_labels;                         /* [B, Q] */
    float *out_scores;                           /* [B, Q] */
    int32_t *counts;                             /* [B] */
} detr_postprocess_desc;
int detr_hip_postprocess(const detr_postprocess_desc *d, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser (detr_tf/optimizers.py:86-88,137-163): per-tensor clip-by-norm then Keras Adam on
 * a flat fp32 parameter buffer.  Tensor t occupies [seg_off[t], seg_off[t+1]) ; chunk c of
 * `chunk` elements belongs to tensor chunk_tensor[c] and starts at chunk_start[c].
 *   hyper (device, float[8]) = {lr_t group0, lr_t group1, lr_t group2, clipnorm, beta1, beta2, eps, -}
 *   with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the host each step.
 *   sumsq [n_chunks]: sumsq_segments stores one partial per chunk; clip_adam sums the partials of a tensor (its chunks are
 *   consecutive) in a fixed order -- no atomics, so data-parallel replicas apply bit-identical updates.
 * ------------------------------------------------------------------------------------------- */
int detr_hip_sumsq_segments_f32(const float *g, const int32_t *chunk_tensor, const int64_t *chunk_start,
                                const int64_t *seg_end, int32_t n_chunks, int32_t chunk, float *sumsq,
                                void *stream);
int detr_hip_clip_adam_f32(float *param, const float *g, float *m, float *v, const int32_t *chunk_tensor,
                           const int64_t *chunk_start, const int64_t *seg_end, const int32_t *tensor_group,
                           const float *sumsq, const float *hyper, int32_t n_chunks, int32_t chunk,
                           void *stream);
/* dst[0..7] = v0..v7: the values travel as kernel arguments, so the host never has to
 * synchronise to update `hyper` (stream ordered, race free) */
int detr_hip_set_floats8_f32(float *dst, float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                             float v7, void *stream);
/* acc[i] += a * g[i] (gradient accumulation optimizers.py:157) */
int detr_hip_axpy_f32(float *acc, const float *g, float a, int64_t n, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Scratch sizing: the number of bytes `workspace` must hold for the call described by the descriptor to take its
 * DETERMINISTIC split-reduction path (0 = the call needs no scratch; negative = malformed descriptor).  Pointers inside the
 * descriptor are not dereferenced; only shapes, split counts, dtypes and (gemm) whether rowsum_a is set are read.  The same
 * value is what a deferred reduction (detr_gemm_desc.defer_out / detr_layernorm_desc.defer_blocks_out) needs as its private slab. */
int64_t detr_hip_workspace_bytes_gemm(const detr_gemm_desc *d);
int64_t detr_hip_workspace_bytes_conv3x3(const detr_conv3x3_desc *d, int32_t mode);
int64_t detr_hip_workspace_bytes_stem(const detr_stem_desc *d, int32_t mode);
int64_t detr_hip_workspace_bytes_layernorm(const detr_layernorm_desc *d);

/* ---------------------------------------------------------------------------------------------
 * Tile plan of the 8-wave LDS-DMA "ring" GEMM (csrc/gemm_ring.h, round 5) that detr_hip_gemm_f32 takes for the tall unsplit
 * bf16 GEMMs (the K >= 512 1x1 convolutions of resnet_backbone.py:119-135 -- forward and input gradient -- and the FFN of
 * transformer.py:172-177): out[0..7] = {32-row blocks per wave, 32-column blocks per wave, ring stages, row pitch of the tile
 * grid, tiles along M, tiles along N, workgroups, LDS bytes}.  Returns 1 when the shape has a plan, 0 when it has none.  Pure
 * host arithmetic (tests, tuning scripts); `out` may be null. */
int detr_hip_gemm_ring_plan(int32_t M, int32_t N, int32_t K, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* DETR_HIP_H */
