// gemm_f32.hip -- generic batched fp32 GEMM with fused epilogue on the MFMA tile engine.
// Replaces every tf.matmul / 1x1 Conv2D / Linear of the reference's hot path and their
// gradients (see include/detr_hip.h for the reference lines).
#include "gemm_core.h"
#include "gemm_bf16_core.h"
#include "gemm_stream.h"

#include "gemm_kernels.h"
#include "gemm_x3.h"
#include "gemm_ring.h"

namespace detr {

// Reduction of the split-K partial slabs: 256 threads = (256 / G) float4 outputs x G split groups (the groups stride the
// split index), combined through LDS in a fixed order.  G = 4 for large outputs; G = 16 for small ones (a 256 x 256
// weight gradient has only 16 K float4 outputs: with G = 4 the launch is 256 blocks and purely latency bound).
struct ReduceOne {
    const float *ws; int splits; long long part_stride; int rows, cols; float *C; long long ldc;
    float alpha; const float *scale; int vec; const float *rs_ws; float *rs_out; float rs_alpha; int nblocks;
    int ts_bm, ts_bn, ts_tn;     // tile-ordered slabs (0: row-major)
};
struct ReduceGroupArgs { ReduceOne r[GEMM_MAX_GROUP]; };

template <int G>
__device__ __forceinline__ void splitk_reduce_body(const float *__restrict__ ws, int splits, long long part_stride,
                                                   int rows, int cols, float *__restrict__ C, long long ldc,
                                                   float alpha, const float *__restrict__ scale, int vec,
                                                   const float *__restrict__ rs_ws, float *__restrict__ rs_out,
                                                   float rs_alpha, const int bid, const int nbid,
                                                   const int ts_bm = 0, const int ts_bn = 0, const int ts_tn = 1) {
    constexpr int OUT = 256 / G;
    __shared__ float4 red[G][OUT];
    if (rs_ws) {     // fused bias gradient: partial row sums [splits][rows] -> rs_out[rows] (fixed summation order)
        for (int m = bid * 256 + threadIdx.x; m < rows; m += nbid * 256) {
            float t = 0.0f;
            for (int k = 0; k < splits; ++k) t += rs_ws[(long long)k * rows + m];
            rs_out[m] += rs_alpha * t;
        }
    }
    const int lo = threadIdx.x % OUT, grp = threadIdx.x / OUT;
    if (ts_bm > 0) {
        // tile-ordered slabs: unit i (a float4 = 4 consecutive ROWS of one column, gemm_core.h: slab_ts_unit) is summed over the
        // splits in the same order as a row-major element would be (k = grp, grp + G, ..., then the G groups in order), so the
        // result is bit-identical to the row-major path.  Reads are lane-linear.  The four lanes of a quad hold a 4 x 4 block
        // (4 rows x 4 consecutive columns): it is transposed through LDS so that every lane updates C with ONE 16-byte
        // read-modify-write of a row (first version: four 4-byte ones per lane -- the reduce launches got 30 % slower).
        __shared__ float tr[OUT][5];
        const int upt = (ts_bm * ts_bn) >> 2;
        const long long total = part_stride >> 2;           // a multiple of 1024: a quad is inside or outside as a whole
        for (long long base = (long long)bid * OUT; base < total; base += (long long)nbid * OUT) {
            const long long i = base + lo;
            const bool valid = i < total;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) {
                const float *p = ws + i * 4;
#pragma unroll 4
                for (int k = grp; k < splits; k += G) {
                    const float4 v = *reinterpret_cast<const float4 *>(p + k * part_stride);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
            red[grp][lo] = s;
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int q = 1; q < G; ++q) {
                    const float4 t = red[q][lo];
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                tr[lo][0] = s.x; tr[lo][1] = s.y; tr[lo][2] = s.z; tr[lo][3] = s.w;
            }
            __syncthreads();
            if (grp == 0 && valid) {
                const int qb = lo & ~3, t = lo & 3;             // lane t of the quad takes row t of the block
                const long long iq = base + qb;
                const int tile = (int)(iq / upt), u = (int)(iq - (long long)tile * upt);
                int row0, col;
                slab_ts_unit(u, ts_bm, ts_bn, row0, col);
                const int row = row0 + (tile / ts_tn) * ts_bm + t;
                col += (tile % ts_tn) * ts_bn;
                if (row < rows && col < cols) {                 // (cols % 4 == 0, col % 4 == 0: the four columns exist together)
                    const float4 v = make_float4(tr[qb][t], tr[qb + 1][t], tr[qb + 2][t], tr[qb + 3][t]);
                    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (scale) sc = *reinterpret_cast<const float4 *>(scale + col);
                    float4 *dst = reinterpret_cast<float4 *>(C + (long long)row * ldc + col);
                    float4 o = *dst;
                    o.x += alpha * sc.x * v.x; o.y += alpha * sc.y * v.y; o.z += alpha * sc.z * v.z; o.w += alpha * sc.w * v.w;
                    *dst = o;
                }
            }
            // (no third barrier: red[] is rewritten after barrier 2, behind every read of it; tr[] is rewritten after the NEXT
            //  iteration's barrier 1, which the lanes reading it here have to reach first)
        }
    } else if (vec) {
        const int c4n = cols >> 2;
        const long long total = (long long)rows * c4n;
        for (long long base = (long long)bid * OUT; base < total; base += (long long)nbid * OUT) {
            const long long i = base + lo;
            const bool valid = i < total;
            int r = 0, c = 0;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) {
                r = (int)(i / c4n);
                c = (int)(i - (long long)r * c4n) * 4;
                const float *p = ws + (long long)r * cols + c;
#pragma unroll 4
                for (int k = grp; k < splits; k += G) {
                    const float4 v = *reinterpret_cast<const float4 *>(p + k * part_stride);
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
            red[grp][lo] = s;
            __syncthreads();
            if (grp == 0 && valid) {
#pragma unroll
                for (int q = 1; q < G; ++q) {
                    const float4 t = red[q][lo];
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
                if (scale) sc = *reinterpret_cast<const float4 *>(scale + c);
                float4 *dst = reinterpret_cast<float4 *>(C + (long long)r * ldc + c);
                float4 o = *dst;
                o.x += alpha * sc.x * s.x; o.y += alpha * sc.y * s.y; o.z += alpha * sc.z * s.z; o.w += alpha * sc.w * s.w;
                *dst = o;
            }
            __syncthreads();
        }
    } else {
        const long long total = (long long)rows * cols;
        for (long long i = bid * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)nbid * blockDim.x) {
            const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += ws[k * part_stride + i];
            C[(long long)r * ldc + c] += alpha * (scale ? scale[c] : 1.0f) * s;
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, int splits, long long part_stride,
                                                            int rows, int cols, float *__restrict__ C, long long ldc,
                                                            float alpha, const float *__restrict__ scale, int vec,
                                                            const float *__restrict__ rs_ws, float *__restrict__ rs_out,
                                                            float rs_alpha, int ts_bm, int ts_bn, int ts_tn) {
    splitk_reduce_body<G>(ws, splits, part_stride, rows, cols, C, ldc, alpha, scale, vec, rs_ws, rs_out, rs_alpha, blockIdx.x,
                          gridDim.x, ts_bm, ts_bn, ts_tn);
}
template <int G>
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(ReduceGroupArgs R) {
    const ReduceOne &r = R.r[blockIdx.y];
    if ((int)blockIdx.x >= r.nblocks) return;
    splitk_reduce_body<G>(r.ws, r.splits, r.part_stride, r.rows, r.cols, r.C, r.ldc, r.alpha, r.scale, r.vec, r.rs_ws, r.rs_out,
                          r.rs_alpha, blockIdx.x, r.nblocks, r.ts_bm, r.ts_bn, r.ts_tn);
}

constexpr int REDUCE_MANY = 16;
struct ReduceManyArgs { ReduceOne r[REDUCE_MANY]; };
template <int G>
__global__ __launch_bounds__(256) void splitk_reduce_many_kernel(ReduceManyArgs R) {
    const ReduceOne &r = R.r[blockIdx.y];
    if ((int)blockIdx.x >= r.nblocks) return;
    splitk_reduce_body<G>(r.ws, r.splits, r.part_stride, r.rows, r.cols, r.C, r.ldc, r.alpha, r.scale, r.vec, r.rs_ws, r.rs_out,
                          r.rs_alpha, blockIdx.x, r.nblocks, r.ts_bm, r.ts_bn, r.ts_tn);
}

// `small` (16 instead of 4 split groups per block) is a function of (rows, cols, splits) alone -- the summation order, and with it
// the bits of the result, must not depend on the slab layout
static void reduce_plan(ReduceOne &r, bool &small) {
    r.vec = (r.cols % 4 == 0) && (r.ldc % 4 == 0) && (r.part_stride % 4 == 0) && aligned16(r.ws) && aligned16(r.C) &&
            (!r.scale || aligned16(r.scale));
    const bool ts = r.ts_bm > 0;
    const bool vec_like = r.vec || ts;
    const long long total = (long long)r.rows * (vec_like ? r.cols / 4 : r.cols);
    small = vec_like && total <= 65536 && r.splits >= 8;
    const long long units = ts ? r.part_stride / 4 : total;
    long long grid = vec_like ? (units + (small ? 15 : 63)) / (small ? 16 : 64) : (units + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    r.nblocks = (int)grid;
}

void launch_splitk_reduce(const float *ws, int splits, long long part_stride, int rows, int cols, float *C, long long ldc,
                          float alpha, const float *scale, hipStream_t stream, const float *rs_ws, float *rs_out,
                          float rs_alpha, int ts_bm, int ts_bn, int ts_tiles_n) {
    ReduceOne r;
    r.ws = ws; r.splits = splits; r.part_stride = part_stride; r.rows = rows; r.cols = cols; r.C = C; r.ldc = ldc;
    r.alpha = alpha; r.scale = scale; r.rs_ws = rs_ws; r.rs_out = rs_out; r.rs_alpha = rs_alpha;
    r.ts_bm = ts_bm; r.ts_bn = ts_bn; r.ts_tn = ts_tiles_n > 0 ? ts_tiles_n : 1;
    bool small;
    reduce_plan(r, small);
    if (small)
        hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3((unsigned)r.nblocks), dim3(256), 0, stream, ws, splits, part_stride, rows, cols,
                           C, ldc, alpha, scale, r.vec, rs_ws, rs_out, rs_alpha, r.ts_bm, r.ts_bn, r.ts_tn);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3((unsigned)r.nblocks), dim3(256), 0, stream, ws, splits, part_stride, rows, cols,
                           C, ldc, alpha, scale, r.vec, rs_ws, rs_out, rs_alpha, r.ts_bm, r.ts_bn, r.ts_tn);
}

template <int BM, int BN, int WGM, int WGN>
static int launch_cfg(const GemmArgs &g, int batch, hipStream_t s, bool ak, bool bk) {
    GemmArgs a = g;
    a.tiles_m = cdiv(g.M, BM);
    a.tiles_n = cdiv(g.N, BN);
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n), 1, (unsigned)(batch * g.split_k));
    dim3 block(GEMM_THREADS);
    if (ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WGM, WGN, true, true>), grid, block, 0, s, a);
    else if (ak && !bk) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WGM, WGN, true, false>), grid, block, 0, s, a);
    else if (!ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WGM, WGN, false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WGM, WGN, false, false>), grid, block, 0, s, a);
    return 0;
}

// compute = 2 on the shapes of gemm_split3_shape: the f32x3 kernel (gemm_x3.h)
template <int BM, int BN>
static int launch_x3(const GemmArgs &g, int batch, hipStream_t s, bool ak, bool bk) {
    GemmArgs a = g;
    a.tiles_m = cdiv(g.M, BM);
    a.tiles_n = cdiv(g.N, BN);
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n), 1, (unsigned)(batch * g.split_k));
    dim3 block(GEMM_THREADS);
    // DETR_HIP_X3_DB: 1 = double-buffered LDS (in-wave overlap); 0 / 2 = single-buffered, which measured faster on every shape of the step
    // (profiles/r06_micro_split3.txt: two workgroups per CU overlap better than one with two buffers)
    const bool db = tune(T_X3_DB) == 1;
    if constexpr (BM == 192) {       // (K-contiguous A only: the transpose-read image of an MN-contiguous operand needs a power-of-two block count)
        if (bk) hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, true, true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, true, false, false>), grid, block, 0, s, a);
    } else {
#define DETR_X3_LAUNCH(DB_)                                                                                              \
    do {                                                                                                                 \
        if (ak && bk) hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, true, true, DB_>), grid, block, 0, s, a);               \
        else if (ak && !bk) hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, true, false, DB_>), grid, block, 0, s, a);        \
        else if (!ak && bk) hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, false, true, DB_>), grid, block, 0, s, a);        \
        else hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, false, false, DB_>), grid, block, 0, s, a);                      \
    } while (0)
        if (db) {
            DETR_X3_LAUNCH(true);
        } else DETR_X3_LAUNCH(false);
    }
#undef DETR_X3_LAUNCH
    return 0;
}

template <int BM, int BN, int WGM, int WGN>
static int launch_cfg_bf16(const GemmArgs &g, int batch, hipStream_t s, bool ak, bool bk, int deep = 0) {
    GemmArgs a = g;
    a.tiles_m = cdiv(g.M, BM);
    a.tiles_n = cdiv(g.N, BN);
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n), 1, (unsigned)(batch * g.split_k));
    dim3 block(GEMM_THREADS);
    if constexpr ((BM == 64 && BN == 64) || (BM == 128 && BN == 128)) {
        if (deep && g.a16 && g.b16) {
            if (ak && bk) hipLaunchKernelGGL((gemm_bf16c_k64_kernel<BM, BN, WGM, WGN, true, true>), grid, block, 0, s, a);
            else if (ak && !bk) hipLaunchKernelGGL((gemm_bf16c_k64_kernel<BM, BN, WGM, WGN, true, false>), grid, block, 0, s, a);
            else if (!ak && bk) hipLaunchKernelGGL((gemm_bf16c_k64_kernel<BM, BN, WGM, WGN, false, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((gemm_bf16c_k64_kernel<BM, BN, WGM, WGN, false, false>), grid, block, 0, s, a);
            return 0;
        }
    }
#define DETR_BF16_LAUNCH(A16_, B16_)                                                                                          \
    do {                                                                                                                          \
        if (ak && bk) hipLaunchKernelGGL((gemm_bf16c_kernel<BM, BN, WGM, WGN, true, true, A16_, B16_>), grid, block, 0, s, a);        \
        else if (ak && !bk) hipLaunchKernelGGL((gemm_bf16c_kernel<BM, BN, WGM, WGN, true, false, A16_, B16_>), grid, block, 0, s, a); \
        else if (!ak && bk) hipLaunchKernelGGL((gemm_bf16c_kernel<BM, BN, WGM, WGN, false, true, A16_, B16_>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((gemm_bf16c_kernel<BM, BN, WGM, WGN, false, false, A16_, B16_>), grid, block, 0, s, a);               \
    } while (0)
    if (g.a16 && g.b16) DETR_BF16_LAUNCH(true, true);
    else if (g.a16) DETR_BF16_LAUNCH(true, false);
    else if (g.b16) DETR_BF16_LAUNCH(false, true);
    else DETR_BF16_LAUNCH(false, false);
#undef DETR_BF16_LAUNCH
    return 0;
}

}  // namespace detr

using namespace detr;

// Host side of one GEMM: validation + kernel arguments (gemm_prepare), then the launch(es) (gemm_launch).  The grouped entry
// point prepares every member the same way and, when they share one kernel variant, issues ONE launch for all of them.
struct GemmPlan {
    GemmArgs g;
    int batch, split;
    bool ak, bk, bf16c, partial;
    bool split3;              // compute = 2: fp32 storage and accuracy on the bf16 matrix pipe (3-way operand split; 64x64 / 128x128 tiles)
    int deep;                 // K-tile depth of the all-bf16 variants: 0 = 32, 1 = 64, 2 = 128 (64x64 tiles only)
    int tile;                 // 0: 64x64, 1: 128x128, 2: 128x64, 3: 128x32, 4: 64x256 (fp32) / 64x128 (bf16), 5: 256x64
    long long part;           // floats per split slab (row-major: M*N; tile-ordered: the padded tile grid)
    int ts_bm, ts_bn, ts_tn;  // tile-ordered slabs (0: row-major)
    EpiArgs final_e;
    const detr_gemm_desc *d;
};

// Tile shape of a GEMM: 0: 64x64, 1: 128x128, 2: 128x64, 3: 128x32, 4: 64x256 (fp32) / 64x128 (bf16), 5: 256x64.  Shared by the
// launch path and by the scratch-size query (the tile-ordered split-K slabs are padded to whole tiles).
// compute = 2 (f32x3): the shapes whose products go through the split kernel (gemm_core.h: mma_ktile_split3).  The operand split is VALU work
// per fragment, so the narrow (N <= 64: 0.77x) and the short-K (K <= 128: 0.92-0.96x) 1x1-convolution GEMMs and the 64-row weight gradients
// (0.85x) keep the exact kernel (scripts/micro_split3.py, profiles/r06_micro_split3.txt) -- both are fp32-accurate, the mode only decides
// which matrix instruction computes the products.  DETR_HIP_SPLIT3_ALL=1: every shape (tests).
static bool gemm_split3_shape(const detr_gemm_desc *d) {
    return d->compute == 2 && d->N > 32 && (tune(T_SPLIT3_ALL) == 1 || (d->N >= 128 && d->K >= 256 && d->M >= 128));
}
static int gemm_pick_tile(const detr_gemm_desc *d, int split, int batch) {
    const bool bf16c = d->compute == 1;
    const int force = tune(T_GEMM_TILE);     // tuning hook (scripts/tune_gemm.py); 0 = heuristic
    int tile = 0;
    if (bf16c) {
        // measured (profiles/tune_bf16_r1c.txt, buffer-descriptor loaders + transpose-read LDS images): the 64x64 tile
        // (7-8 waves/SIMD) wins every non-split shape and the short reductions; 128x128 (3 waves/SIMD, 4x the operand
        // reuse) pays only for the long split-K weight gradients (K >= 16384, N >= 128) and for unsplit K >= 1024 GEMMs
        // that still fill the chip with 128x128 tiles (M33600 N256 K1024: 54 vs 60 us)
        const long long t128 = (long long)cdiv(d->M, 128) * cdiv(d->N, 128) * batch;
        // round 3, cold-cache sweeps (scripts/micro_wgrad.py, micro_gemm.py --cold; profiles/r03_micro_*): the layer4 weight
        // gradients (M, N >= 512, K = 8400) gain 15-25 % on 128x128 tiles (512x2048: 64 -> 53 us, 1024x2048: 109 -> 80), and so do
        // the wide K = 512 GEMMs of layer3's first block (M33600 N1024: 100 -> 89 us, with residual + mask 163 -> 151)
        // round 4 (scripts/experiments/tile_force.sh): the encoder's FFN weight-gradient pair (256 x 2048 and 2048 x 256 outputs, K = 8400)
        // 54.7 us as one grouped launch of 64x64 tiles -> 45.3 us as two launches of 128x128 tiles
        const bool big_split = (d->N >= 128 && d->K >= 16384) || (d->M >= 512 && d->N >= 512 && d->K >= 4096) ||
                               (d->K >= 4096 && ((d->M >= 256 && d->N >= 2048) || (d->M >= 2048 && d->N >= 256)));
        // round 4 (scripts/experiments/tile_shapes.py, every unsplit shape of the step on every tile): the rules hold within +-5 % except
        // M133600 N256 K512 (layer2's projection shortcut and its input gradient): 81.9 us on 64x64 tiles, 71.1 on 128x128
        const bool big_plain = (d->K >= 1024 && t128 >= 512) || (d->K >= 512 && d->N >= 512 && t128 >= 1024) ||
                               (d->K >= 512 && d->N >= 256 && t128 >= 2048);
        const bool small = (split > 1) ? !big_split : !big_plain;
        if (force == 2) tile = 2;
        else if (force == 5) tile = 4;
        else if (force == 3 || (force == 0 && small)) tile = 0;
        else if (force == 0 && split > 1 && d->M <= 64) tile = 4;     // 64 output rows: half of a 128-row tile would be padding (M64 N256 K534400: 84 -> 74 us)
        else tile = 1;
        // ring weight gradient (gemm_ring.h) on wider tiles -- experiment hook DETR_HIP_RING_WTILE: 1 = 128 x 256, 2 = 256 x 128
        // (the FULL predicate of the ring weight gradient in gemm_launch -- ADVICE r5: a tile only that kernel has must not be handed to a
        //  launch the kernel will not take, nor size the slab scratch for it; the slab form itself is checked again there)
        if (tile == 1 && split > 1 && batch == 1 && d->a_dtype == 1 && d->b_dtype == 1 && !d->a_kcontig && !d->b_kcontig && !d->rowsum_a &&
            tune(T_GEMM_RING) != 2 && tune(T_GEMM_RING) != 3 && tune(T_SLAB_TS) != 2 && d->M % 8 == 0 && d->N % 8 == 0 && d->lda % 8 == 0 &&
            d->ldb % 8 == 0 && aligned16(d->A) && aligned16(d->B) && aligned16(d->C) && d->ldc % 4 == 0 && (!d->scale || aligned16(d->scale))) {
            const int wt = tune(T_RING_WTILE);
            if (wt == 1 && d->N >= 256) tile = 6;
            else if (wt == 2 && d->M >= 256) tile = 7;
        }
    } else if (gemm_split3_shape(d)) {
        // f32x3 (mma_ktile_split3): a wave's operand split costs VALU time per FRAGMENT, its 6 MFMAs per fragment PAIR -- 64 x 64 wave
        // tiles (128 x 128 workgroup tiles) are matrix-pipe bound where 32 x 32 ones are VALU bound; small grids keep the 64 x 64 tile
        const long long t128 = (long long)cdiv(d->M, 128) * cdiv(d->N, 128) * batch * split;
        const int lim = tune(T_SPLIT3_T128) > 0 ? tune(T_SPLIT3_T128) : 192;
        if (force == 1 || (force == 0 && t128 >= lim)) tile = 1;
        else if (force == 2) tile = 2;        // (128 x 64: three workgroups per CU)
        else tile = 0;
        // Tile-count quantisation (profiles/r06_ab_results.txt #9): two 128 x 128 workgroups fit a CU = 512 slots, and M33600 N256 is 526 tiles -- two
        // rounds, the second with 14 workgroups.  192 x 128 tiles (tile id 8: 1.5 x the work per workgroup, the same two per CU; K-contiguous A,
        // unsplit, no fused row sums) take the GEMM when they need fewer rounds x 1.5.  DETR_HIP_X3_T192 = 2: never, 1: wherever eligible.
        if (tile == 1 && force == 0 && split == 1 && batch == 1 && d->a_kcontig && !d->rowsum_a && tune(T_X3_T192) != 2) {
            const long long slots = 512;
            const long long t192 = (long long)cdiv(d->M, 192) * cdiv(d->N, 128);
            const long long r128 = (t128 + slots - 1) / slots, r192 = (t192 + slots - 1) / slots;
            if (tune(T_X3_T192) == 1 || 3 * r192 < 2 * r128) tile = 8;
        }
        // 128 x 64 tiles (three workgroups per CU = 768 slots) where the 128 x 128 grid is just over a whole number of rounds (M8400 N2048: 1056 tiles = 2.06
        // rounds: 142 -> 135 us, 88.7 -> 77.9 us at K = 256) or a short-K grid fills half the chip (M8400 N512 K256: 34.7 -> 30.4 us)
        if (tile == 1 && force == 0 && split == 1 && batch == 1 && tune(T_X3_T192) != 2 && d->M >= 4096) {
            const long long over = t128 % 512;
            if ((t128 > 512 && over > 0 && over <= 128) || (t128 <= 384 && d->K <= 512)) tile = 2;
        }
    } else if (force == 1) tile = 1;
    else if (force == 2) tile = 2;
    else if (force == 3) tile = 0;
    else if (force == 4) tile = 3;
    else if (force == 5) tile = 4;
    else if (force == 6) tile = 5;
    else if (d->N <= 32) tile = 3;
    else tile = 0;      // measured (profiles/tune_r1.txt): 64x64 (8 waves/SIMD) beats 128x64 by 2-10 % and 128x128 by 15-50 % in fp32
    return tile;
}
// BM x BN of a tile id; ts_ok: the kernel runs a 2 x 2 wave grid (store_slab_ts / slab_ts_unit)
static void gemm_tile_dims(bool bf16c, int tile, int &bm, int &bn, bool &ts_ok) {
    ts_ok = true;
    if (tile == 1) { bm = 128; bn = 128; }
    else if (tile == 8 && !bf16c) { bm = 192; bn = 128; }         // (f32x3, unsplit only)
    else if (tile == 6 && bf16c) { bm = 128; bn = 256; }         // (ring weight gradient only)
    else if (tile == 7 && bf16c) { bm = 256; bn = 128; }
    else if (tile == 2) { bm = 128; bn = 64; }
    else if (tile == 4 && bf16c) { bm = 64; bn = 128; }
    else if (tile == 0) { bm = 64; bn = 64; }
    else { bm = bn = 0; ts_ok = false; }          // fp32 128x32 / 64x256 / 256x64: 4x1 / 1x4 wave grids keep row-major slabs
}
// Split-K slab layout: tile-ordered when the reduce launch can keep its 16-byte form (the same conditions as its row-major
// `vec` path: the summation order -- the bits of the result -- must not depend on the layout).  DETR_HIP_SLAB_TS=2: row-major.
static bool gemm_slab_ts(const detr_gemm_desc *d, int split, int batch, int tile, int &bm, int &bn) {
    bool ok;
    gemm_tile_dims(d->compute == 1, tile, bm, bn, ok);
    return ok && split > 1 && batch == 1 && d->N % 4 == 0 && d->ldc % 4 == 0 && aligned16(d->C) && (!d->scale || aligned16(d->scale)) &&
           tune(T_SLAB_TS) != 2;
}

// split count a GEMM really runs with: no empty splits (recomputed from the K tiles each split gets)
static int gemm_effective_split(const detr_gemm_desc *d) {
    int split = d->split_k > 1 ? d->split_k : 1;
    if (split > 1) {
        const int nkt = cdiv(d->K, (d->compute == 1 || gemm_split3_shape(d)) ? BF_BK : GEMM_BK);      // (the f32x3 kernel walks 32-deep K tiles)
        const int per = cdiv(nkt, split);
        split = cdiv(nkt, per);
    }
    return split;
}

extern "C" int64_t detr_hip_workspace_bytes_gemm(const detr_gemm_desc *d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return -1;
    const int batch = d->batch > 0 ? d->batch : 1;
    const int split = gemm_effective_split(d);
    if (split <= 1 || batch != 1) return 0;          // (batched split-K accumulates with atomics: no scratch)
    int bm, bn;
    int64_t part = (int64_t)d->M * d->N;
    if (gemm_slab_ts(d, split, batch, gemm_pick_tile(d, split, batch), bm, bn))      // tile-ordered slabs: whole tiles
        part = (int64_t)cdiv(d->M, bm) * cdiv(d->N, bn) * bm * bn;
    return (int64_t)split * (part + (d->rowsum_a ? d->M : 0)) * 4;
}

static int gemm_prepare(const detr_gemm_desc *d, GemmPlan &p) {
    DETR_REQUIRE(d != nullptr, "gemm: null descriptor");
    DETR_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    DETR_REQUIRE(d->A && d->B && d->C, "gemm: null operand");
    const int batch = d->batch > 0 ? d->batch : 1;
    const int inner = d->batch_inner > 0 ? d->batch_inner : 1;
    const bool bf16c = d->compute == 1;
    int split = gemm_effective_split(d);
    if (split > 1) {
        DETR_REQUIRE(!d->bias && !d->residual && !d->mask && d->act == 0,
                     "gemm: split_k allows only scale/alpha in the epilogue");
    }
    {   // operands are addressed through 32-bit buffer offsets (gemm_core.h BufSrc)
        const long long ea = d->a_kcontig ? (long long)(d->M - 1) * d->lda + d->K : (long long)(d->K - 1) * d->lda + d->M;
        const long long eb = d->b_kcontig ? (long long)(d->N - 1) * d->ldb + d->K : (long long)(d->K - 1) * d->ldb + d->N;
        DETR_REQUIRE(d->b_dtype == 0 || (d->b_dtype == 1 && bf16c && batch == 1 && aligned16(d->B) && d->ldb % 8 == 0 &&
                                         (d->b_kcontig ? d->K % 8 == 0 : d->N % 4 == 0)),
                     "gemm: a bf16 B operand needs compute = bf16, batch 1, 16-byte alignment, ldb %% 8 == 0 and K %% 8 (N %% 4) == 0");
        DETR_REQUIRE(ea * 4 <= BUF_MAX_BYTES && eb * 4 <= BUF_MAX_BYTES, "gemm: an operand spans more than 4 GB");
        DETR_REQUIRE(d->a_dtype == 0 || (d->a_dtype == 1 && bf16c && batch == 1 && ((uintptr_t)d->A % 16 == 0) &&
                                         (d->a_kcontig ? (d->lda % 8 == 0 && d->K % 8 == 0) : (d->lda % 4 == 0 && d->M % 4 == 0))),
                     "gemm: a bf16 A operand needs compute = bf16, batch 1, 16-byte alignment and K %% 8 (M %% 4) == 0");
        DETR_REQUIRE((d->c_dtype == 0 && d->r_dtype == 0 && d->m_dtype == 0) || (batch == 1 && split == 1),
                     "gemm: bf16 C / residual / mask need batch 1 and no split-K");
    }
    if (d->rowsum_a) DETR_REQUIRE(!d->a_kcontig && batch == 1, "gemm: rowsum_a needs an MN-contiguous A operand and batch == 1");
    DETR_REQUIRE((long long)batch * split <= 65535, "gemm: batch*split_k=%lld exceeds grid.z", (long long)batch * split);

    GemmArgs &g = p.g;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.A = d->A; g.lda = d->lda;
    g.B = d->B; g.ldb = d->ldb;
    g.C = d->C; g.ldc = d->ldc;
    g.batch_inner = inner;
    g.sA0 = d->sA0; g.sA1 = d->sA1; g.sB0 = d->sB0; g.sB1 = d->sB1; g.sC0 = d->sC0; g.sC1 = d->sC1;
    g.split_k = split;
    g.part_stride = 0;
    g.tiles_m = g.tiles_n = 0;
    auto strides_ok = [](long long ld, long long s0, long long s1) { return (ld % 4 == 0) && (s0 % 4 == 0) && (s1 % 4 == 0); };
    // vec: float4 tile loads; the extent along the contiguous axis must be a multiple of 4 so that a float4 is
    // entirely inside or outside the operand (branch-free guarded loads, gemm_core.h ld4_sel)
    g.a_vec = aligned16(d->A) && strides_ok(d->lda, d->sA0, d->sA1) && ((d->a_kcontig ? d->K : d->M) % 4 == 0);
    g.b_vec = aligned16(d->B) && strides_ok(d->ldb, d->sB0, d->sB1) && ((d->b_kcontig ? d->K : d->N) % 4 == 0);
    g.e.alpha = d->alpha;
    g.e.scale = d->scale;
    g.e.bias = d->bias;
    g.e.residual = d->residual; g.e.ldr = d->ldr;
    g.e.mask = d->mask; g.e.ldmask = d->ldmask;
    g.e.act = d->act;
    g.e.atomic = split > 1 ? 1 : 0;
    g.e.drop_scale = 0.0f; g.e.drop_thresh = 0; g.e.drop_seed = 0;
    if (d->dropout_p > 0.0f) {
        DETR_REQUIRE(d->dropout_p < 1.0f && split == 1 && batch == 1, "gemm: dropout needs 0<p<1, no split-K, no batch");
        g.e.drop_scale = 1.0f / (1.0f - d->dropout_p);
        g.e.drop_thresh = drop_thresh16(d->dropout_p);
        g.e.drop_seed = d->dropout_seed;
        g.e.drop_step = d->dropout_step;
    }
    g.e.c16 = d->c_dtype == 1; g.e.r16 = d->r_dtype == 1;
    g.e.m16 = (d->m_dtype == 1 || d->m_dtype == 2) ? d->m_dtype : 0;       // 2: bit-packed mask (one byte per 8 columns, ldmask in bytes)
    g.e.mbits_out = d->maskbits_out; g.e.ld_mbits_out = d->ld_maskbits_out;
    const bool bits_in = d->mask && d->m_dtype == 2, bits_out = d->maskbits_out != nullptr;
    g.e.vec = aligned16(d->C) && (d->ldc % 4 == 0) && (d->sC0 % 4 == 0) && (d->sC1 % 4 == 0) &&
              (!d->scale || aligned16(d->scale)) && (!d->bias || aligned16(d->bias)) &&
              (!d->residual || (aligned16(d->residual) && d->ldr % 4 == 0)) &&
              (!d->mask || bits_in || (aligned16(d->mask) && d->ldmask % 4 == 0));

    // all-bf16 epilogue streams, 16-byte rows: the 8-columns-per-lane epilogue (gemm_core.h epilogue_wide16); DETR_HIP_EPI_WIDE=2: off
    g.e.wide16 = g.e.c16 && split == 1 && batch == 1 && (!d->residual || g.e.r16) && (!d->mask || g.e.m16) && d->N % 8 == 0 &&
                 d->ldc % 8 == 0 && aligned16(d->C) && (!d->residual || (d->ldr % 8 == 0 && aligned16(d->residual))) &&
                 (!d->mask || bits_in || (d->ldmask % 8 == 0 && aligned16(d->mask))) && (!d->scale || aligned16(d->scale)) &&
                 (!d->bias || aligned16(d->bias)) && (tune(T_EPI_WIDE) != 2 || bits_in || bits_out);
    // bit-packed masks live in the all-bf16 epilogues (8 columns per lane = one byte) of the tile engine and of the streaming kernel
    DETR_REQUIRE(!(bits_in || bits_out) || (bf16c && g.e.wide16 && (!bits_in || d->ldmask >= d->N / 8) && (!bits_out || d->ld_maskbits_out >= d->N / 8)),
                 "gemm: bit-packed masks need bf16 C / residual, N %% 8 == 0, 16-byte aligned rows, no split-K / batch, and a row pitch >= N / 8 bytes");

    const bool ak = d->a_kcontig != 0, bk = d->b_kcontig != 0;
    DETR_REQUIRE(!(bf16c && d->a_dtype == 1 && !ak) || (d->lda % 8 == 0 && d->M % 8 == 0 && aligned16(d->A)),
                 "gemm: an M-contiguous bf16 A operand needs lda %% 8 == 0, M %% 8 == 0 and a 16-byte aligned base (16-byte requests)");
    DETR_REQUIRE(!(bf16c && d->b_dtype == 1 && !bk) || (d->ldb % 8 == 0 && d->N % 8 == 0 && aligned16(d->B)),
                 "gemm: an N-contiguous bf16 B operand needs ldb %% 8 == 0, N %% 8 == 0 and a 16-byte aligned base (16-byte requests)");
    const int tile = gemm_pick_tile(d, split, batch);
    // deterministic split-K: partial slabs in the caller's workspace.  Tile-ordered slabs (whole tiles: a little more scratch than
    // split*M*N -- detr_hip_workspace_bytes_gemm reports it) when the workspace holds them, row-major slabs when it only holds
    // the documented split*M*N floats, fp32 atomics without a workspace.
    long long part = (long long)d->M * d->N;
    int ts_bm = 0, ts_bn = 0;
    const long long rs_extra = d->rowsum_a ? d->M : 0;
    bool ts = gemm_slab_ts(d, split, batch, tile, ts_bm, ts_bn);
    const long long part_ts = ts ? (long long)cdiv(d->M, ts_bm) * cdiv(d->N, ts_bn) * ts_bm * ts_bn : 0;
    const bool ws_ok = split > 1 && batch == 1 && d->workspace && aligned16(d->workspace);
    if (ts && !(ws_ok && d->workspace_bytes >= (long long)split * (part_ts + rs_extra) * 4)) ts = false;
    if (ts) part = part_ts;
    const bool partial = ws_ok && d->workspace_bytes >= (long long)split * (part + rs_extra) * 4;
    DETR_REQUIRE(!(d->rowsum_a && split > 1 && !partial), "gemm: rowsum_a with split_k needs the workspace path");
    g.rowsum = d->rowsum_a;
    g.rowsum_alpha = d->rowsum_alpha;
    g.rowsum_partial = 0;
    g.b16 = d->b_dtype == 1;
    g.a16 = d->a_dtype == 1;
    g.split_xcd = tune(T_SPLIT_XCD) != 2;
    g.slab_ts = 0;
    p.ts_bm = p.ts_bn = 0; p.ts_tn = 1;
    EpiArgs final_e = g.e;
    if (partial) {      // plain stores of the partial tiles, reduced by a second launch
        g.C = d->workspace;
        g.ldc = d->N;
        g.part_stride = part;
        g.e.alpha = 1.0f;
        g.e.scale = nullptr;
        g.e.atomic = 0;
        g.e.vec = (d->N % 4 == 0);
        if (ts) {
            g.slab_ts = 1;
            p.ts_bm = ts_bm; p.ts_bn = ts_bn; p.ts_tn = cdiv(d->N, ts_bn);
        }
        if (d->rowsum_a) {       // partial row sums go behind the tile slabs
            g.rowsum = d->workspace + (long long)split * part;
            g.rowsum_partial = 1;
        }
    }
    // 64-deep K tiles (all-bf16 operands, tiles 0 / 1): DETR_HIP_GEMM_K64 = 0 rule below, 1 every eligible GEMM, 2 never.
    // Measured (scripts/micro_gemm.py, profiles/r03_micro_gemm_k64.txt): 64x64 tiles gain 10-23 % from K >= 512 per split on (M8400
    // N512 K2048: 47.0 -> 39.2 us, M800 N256 K2048 cold: 28.0 -> 21.6); UNSPLIT 128x128 tiles lose 20-35 % (72 KB of LDS: two
    // workgroups per CU instead of three -- M33600 N1024 K512: 73 -> 92 us), split ones gain 0-7 % (cold: 82.0 -> 76.1 us)
    {
        const int k64 = tune(T_GEMM_K64);
        const int per_split = cdiv(d->K, split);
        p.deep = (bf16c && g.a16 && g.b16 && (tile == 0 || tile == 1) && k64 != 2 &&
                  (k64 == 1 || (per_split >= 512 && (tile == 0 || split > 1)))) ? 1 : 0;
    }
    p.batch = batch; p.split = split; p.ak = ak; p.bk = bk; p.bf16c = bf16c; p.partial = partial; p.tile = tile;
    p.split3 = gemm_split3_shape(d) && (tile == 0 || tile == 1 || tile == 2 || tile == 8);
    p.part = part; p.final_e = final_e; p.d = d;
    return 0;
}

// The streaming kernel (gemm_stream.h) takes the short-K, tall GEMMs whose operands, output, residual and mask are all
// bf16 in memory: the layer1 / layer2 1x1 convolutions and their input gradients.  DETR_HIP_GEMM_STREAM=2 disables it.
static bool gemm_stream_eligible(const GemmPlan &p) {
    const detr_gemm_desc *d = p.d;
    const GemmArgs &g = p.g;
    if (!(p.bf16c && g.a16 && g.b16 && g.e.c16 && p.ak && p.batch == 1 && p.split == 1 && !g.rowsum)) return false;
    // M >= 16384: the backbone's 1x1 convolutions; the transformer's K = 256 FFN GEMMs (M = B*L = 8400: linear1 forward with its
    // fused dropout, the input gradient of linear2 with alpha = 1/(1-p) and the ReLU / dropout mask) take it from M >= 4096
    const bool ext = d->alpha != 1.0f || d->dropout_p > 0.0f;
    if (!((d->K == 64 || d->K == 128 || d->K == 256) && d->N % 64 == 0 && d->M >= ((d->K == 256 && d->N >= 1024) ? 4096 : 16384))) return false;
    if (d->scale || (ext && d->K != 256) || !(d->act == 0 || d->act == 1)) return false;
    if (d->residual && !(g.e.r16 && d->ldr % 8 == 0 && aligned16(d->residual))) return false;
    if (d->mask && g.e.m16 != 2 && !(g.e.m16 && d->ldmask % 8 == 0 && aligned16(d->mask))) return false;
    if (!(d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 8 == 0 && aligned16(d->A) && aligned16(d->B) && aligned16(d->C))) return false;
    const long long ldm_eff = g.e.m16 == 2 ? 0 : d->ldmask;
    long long pitch = d->ldr > ldm_eff ? d->ldr : ldm_eff;
    if (d->ldc > pitch) pitch = d->ldc;             // C is stored through a buffer descriptor as well (32-bit byte offsets)
    const long long span = (long long)(d->M - 1) * pitch + d->N;
    if (span * 2 > BUF_MAX_BYTES) return false;
    // mask bits OUT: the kernel is instantiated for the form that produces them in the step (block outputs: [k][n] weights + residual)
    if (d->maskbits_out && (p.bk || !d->residual || d->mask || ext)) return false;
    return tune(T_GEMM_STREAM) != 2;
}

static void gemm_stream_launch(const GemmPlan &p, hipStream_t s) {
    const detr_gemm_desc *d = p.d;
    StreamArgs a;
    a.M = d->M; a.N = d->N;
    a.A = reinterpret_cast<const unsigned short *>(d->A); a.lda = d->lda;
    a.B = reinterpret_cast<const unsigned short *>(d->B); a.ldb = d->ldb;
    a.C = reinterpret_cast<unsigned short *>(d->C); a.ldc = d->ldc;
    a.res = reinterpret_cast<const unsigned short *>(d->residual); a.ldr = d->ldr;
    a.mask = reinterpret_cast<const unsigned short *>(d->mask); a.ldm = d->ldmask;
    a.mbits_out = d->maskbits_out; a.ld_mbits = d->ld_maskbits_out;
    a.bias = d->bias;
    a.act = d->act;
    a.alpha = d->alpha;
    a.drop_scale = p.g.e.drop_scale; a.drop_thresh = p.g.e.drop_thresh; a.drop_seed = p.g.e.drop_seed; a.drop_step = p.g.e.drop_step;
    a.n_tiles = a.row_tiles = a.q = 0;
    const bool mb = d->mask && d->m_dtype == 2;
    if (d->K == 64) launch_gemm_stream<64>(a, p.bk, s, mb);
    else if (d->K == 128) launch_gemm_stream<128>(a, p.bk, s, mb);
    else launch_gemm_stream<256>(a, p.bk, s, mb);
}

// GEMM with the LayerNorm of its output rows fused (detr_gemm_desc.ln_y; gemm_core.h: epilogue_ln): row-complete 32 x 256 tiles
static int gemm_ln_launch(const GemmPlan &p, hipStream_t s) {
    const detr_gemm_desc *d = p.d;
    const GemmArgs &g0 = p.g;
    DETR_REQUIRE(p.bf16c && d->N == LN_TILE_N && d->ldc == LN_TILE_N && p.ak && p.bk && g0.b16 && p.batch == 1 && p.split == 1,
                 "gemm + LayerNorm: needs compute = 1, N = ldc = 256, k-contiguous A and bf16 k-contiguous B, no batch / split_k");
    DETR_REQUIRE(d->act == 0 && !d->scale && !d->mask && !g0.e.c16 && !g0.e.r16 && !d->rowsum_a && !d->maskbits_out && d->K % 8 == 0,
                 "gemm + LayerNorm: epilogue is bias, alpha, dropout and an fp32 residual only (K %% 8 == 0)");
    DETR_REQUIRE(d->ln_gamma && d->ln_beta && d->ln_mean && d->ln_rstd && (!d->ln_y2 || (d->ln_add && d->ln_add_rows > 0)),
                 "gemm + LayerNorm: gamma, beta, mean, rstd are required (and add / add_rows with y2)");
    DETR_REQUIRE((!d->residual || (d->ldr % 4 == 0 && aligned16(d->residual))) && aligned16(d->C) && aligned16(d->ln_y) &&
                 (!d->bias || aligned16(d->bias)) && aligned16(d->ln_gamma) && aligned16(d->ln_beta) &&
                 (!d->ln_y2 || (aligned16(d->ln_y2) && aligned16(d->ln_add))) && (!d->ln_y16 || aligned16(d->ln_y16)),
                 "gemm + LayerNorm: rows must be 16-byte aligned");
    GemmArgs a = g0;
    a.tiles_m = cdiv(d->M, LN_TILE_M);
    a.tiles_n = 1;
    a.ln.gamma = d->ln_gamma; a.ln.beta = d->ln_beta; a.ln.y = d->ln_y; a.ln.mean = d->ln_mean; a.ln.rstd = d->ln_rstd;
    a.ln.add = d->ln_add; a.ln.add_rows = d->ln_add_rows; a.ln.y2 = d->ln_y2; a.ln.y16 = d->ln_y16; a.ln.eps = d->ln_eps;
    const dim3 grid((unsigned)a.tiles_m), block(GEMM_THREADS);
    if (g0.a16) hipLaunchKernelGGL(gemm_bf16c_ln_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(gemm_bf16c_ln_kernel<false>, grid, block, 0, s, a);
    DETR_LAUNCH_CHECK("gemm + LayerNorm");
    if (d->defer_out) d->defer_out->splits = 0;
    return 0;
}

// The ring kernel (gemm_ring.h) takes the tall unsplit GEMMs whose operands are both bf16 in memory with a K-contiguous A: the
// K >= 512 1x1 convolutions of layer2-layer4 (forward and input gradient) and the encoder's FFN at M = B*L.  DETR_HIP_GEMM_RING=2
// disables it, =1 takes every eligible shape (tests: small M as well).
static bool gemm_ring_eligible(const GemmPlan &p, RingPlan &rp) {
    const detr_gemm_desc *d = p.d;
    const GemmArgs &g = p.g;
    const int mode = tune(T_GEMM_RING);
    if (mode == 2) return false;
    if (!(p.bf16c && g.a16 && g.b16 && p.ak && p.batch == 1 && p.split == 1 && !g.rowsum && !d->ln_y)) return false;
    if (!(d->K % RING_BK == 0 && d->K >= 2 * RING_BK && d->N >= 128 && d->N % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && aligned16(d->A) &&
          aligned16(d->B))) return false;
    // which shapes (scripts/micro_gemm.py / micro_ring.py, profiles/r05_micro_gemm_ring.txt): every tall K >= 512 GEMM (-15 .. -30 % against the
    // 4-wave engine); the decoder's 800-row K = 2048 FFN GEMMs (15.2 -> 11.0 us); and, in front of the streaming kernel, two K = 256
    // families it runs faster than that kernel does: layer3's N = 1024 block outputs / input gradients (M33600: 51.5 -> 45.7,
    // 62.5 -> 51.8 us) and the encoder FFN's input gradient with [k][n] weights (M8400 N2048: 31.6 -> 26.6 us)
    const bool tall = d->M >= 4096 && d->K >= 512;
    const bool dec_ffn = d->M >= 512 && d->K >= 1024;
    const bool k256 = d->K == 256 && d->N >= 1024 && (d->M >= 16384 || (d->M >= 4096 && !p.bk));
    if (mode != 1 && !(tall || dec_ffn || k256)) return false;
    return gemm_ring_plan(d->M, d->N, d->K, rp);
}

// fp32 form (the exact-f32 parity mode): fp32 operands, K-contiguous A, unsplit; DETR_HIP_GEMM_RING as above
static bool gemm_ring_f32_eligible(const GemmPlan &p, RingPlan &rp) {
    const detr_gemm_desc *d = p.d;
    const GemmArgs &g = p.g;
    const int mode = tune(T_GEMM_RING);
    if (mode == 2) return false;
    if (p.bf16c || p.split3 || g.a16 || g.b16 || !p.ak || p.batch != 1 || p.split != 1 || g.rowsum || d->ln_y) return false;
    if (d->c_dtype || d->r_dtype || d->m_dtype) return false;
    if (!(d->K % 32 == 0 && d->K >= 64 && d->N >= 128 && d->lda % 4 == 0 && d->ldb % 4 == 0 && aligned16(d->A) && aligned16(d->B) &&
          (p.bk || d->N % 4 == 0))) return false;
    // measured (scripts/micro_ring.py, RING_F32=1, second-pass columns -- the first timings of a process run at a lower clock): -3 .. -9 % on
    // the backbone's M >= 33600 shapes, +-0 .. +3 % on the M = 8400 ones: the fp32 MFMA kernels sit at ~0.6 of the nominal peak whatever
    // the structure (NOTEBOOK 4c)
    if (mode != 1 && !(d->M >= 16384)) return false;
    return gemm_ring_f32_plan(d->M, d->N, d->K, rp);
}

static int gemm_launch(const GemmPlan &p, hipStream_t s) {
    const GemmArgs &g = p.g;
    const detr_gemm_desc *d = p.d;
    const int batch = p.batch;
    const bool ak = p.ak, bk = p.bk;
    if (d->ln_y) return gemm_ln_launch(p, s);
    {
        RingPlan rp;
        const bool ring_first = tune(T_GEMM_STREAM) != 3 && gemm_ring_eligible(p, rp);      // (DETR_HIP_GEMM_STREAM=3: the streaming kernel keeps its K = 256 shapes)
        if (!ring_first && gemm_stream_eligible(p)) {
            gemm_stream_launch(p, s);
            DETR_LAUNCH_CHECK("gemm (stream)");
            return 0;
        }
        if (ring_first) {
            if (gemm_ring_launch(g, bk, rp, s)) return -1;
            DETR_LAUNCH_CHECK("gemm (ring)");
            if (d->defer_out) d->defer_out->splits = 0;
            return 0;
        }
        if (gemm_ring_f32_eligible(p, rp)) {
            if (gemm_ring_f32_launch(g, bk, rp, s)) return -1;
            DETR_LAUNCH_CHECK("gemm (ring, fp32)");
            if (d->defer_out) d->defer_out->splits = 0;
            return 0;
        }
    }
    // split-K weight gradients of the backbone's 1x1 convolutions on 128 x 128 tiles: the 8-wave ring form (gemm_ring.h; bit-identical slabs)
    const bool ring_wgrad = p.bf16c && (p.tile == 1 || p.tile == 6 || p.tile == 7) && g.a16 && g.b16 && !ak && !bk && batch == 1 && p.split > 1 &&
                            g.slab_ts && !g.rowsum && d->M % 8 == 0 && d->N % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && aligned16(d->A) &&
                            aligned16(d->B) && tune(T_GEMM_RING) != 2 && tune(T_GEMM_RING) != 3;
    DETR_REQUIRE(ring_wgrad || (p.tile != 6 && p.tile != 7), "gemm: the 128 x 256 / 256 x 128 tiles exist for the ring weight gradient only "
                 "(tile-ordered slabs in a workspace of detr_hip_workspace_bytes_gemm bytes are required)");
    if (ring_wgrad) {
        if (gemm_ring_wgrad_launch(g, p.ts_bm, p.ts_bn, p.split, s)) return -1;
    } else if (p.bf16c) {
        if (p.tile == 2) launch_cfg_bf16<128, 64, 2, 2>(g, batch, s, ak, bk);
        else if (p.tile == 4) launch_cfg_bf16<64, 128, 2, 2>(g, batch, s, ak, bk);
        else if (p.tile == 0) launch_cfg_bf16<64, 64, 2, 2>(g, batch, s, ak, bk, p.deep);
        else launch_cfg_bf16<128, 128, 2, 2>(g, batch, s, ak, bk, p.deep);
    } else if (p.split3 && p.tile == 8) launch_x3<192, 128>(g, batch, s, ak, bk);
    else if (p.split3 && p.tile == 1) launch_x3<128, 128>(g, batch, s, ak, bk);
    else if (p.split3 && p.tile == 2) launch_x3<128, 64>(g, batch, s, ak, bk);
    else if (p.split3) launch_x3<64, 64>(g, batch, s, ak, bk);
    else if (p.tile == 1) launch_cfg<128, 128, 2, 2>(g, batch, s, ak, bk);
    else if (p.tile == 2) launch_cfg<128, 64, 2, 2>(g, batch, s, ak, bk);
    else if (p.tile == 3) launch_cfg<128, 32, 4, 1>(g, batch, s, ak, bk);
    else if (p.tile == 4) launch_cfg<64, 256, 1, 4>(g, batch, s, ak, bk);
    else if (p.tile == 5) launch_cfg<256, 64, 4, 1>(g, batch, s, ak, bk);
    else launch_cfg<64, 64, 2, 2>(g, batch, s, ak, bk);
    DETR_LAUNCH_CHECK("gemm");
    if (p.partial && d->defer_out) {      // the caller reduces later, many slabs per launch (detr_hip_splitk_reduce_many)
        detr_reduce_desc *o = d->defer_out;
        o->ws = d->workspace; o->splits = p.split; o->part_stride = p.part; o->rows = d->M; o->cols = d->N;
        o->C = d->C; o->ldc = d->ldc; o->alpha = p.final_e.alpha; o->scale = p.final_e.scale;
        o->rs_ws = d->rowsum_a ? d->workspace + (long long)p.split * p.part : nullptr;
        o->rs_out = d->rowsum_a; o->rs_alpha = d->rowsum_alpha;
        o->ts_bm = p.ts_bm; o->ts_bn = p.ts_bn; o->ts_tiles_n = p.ts_tn;
        return 0;
    }
    if (d->defer_out) d->defer_out->splits = 0;
    if (p.partial) {
        launch_splitk_reduce(d->workspace, p.split, p.part, d->M, d->N, d->C, d->ldc, p.final_e.alpha, p.final_e.scale, s,
                             d->rowsum_a ? d->workspace + (long long)p.split * p.part : nullptr, d->rowsum_a, d->rowsum_alpha,
                             p.ts_bm, p.ts_bn, p.ts_tn);
        DETR_LAUNCH_CHECK("gemm split-k reduce");
    }
    return 0;
}

// Which kernel family detr_hip_gemm_f32 would launch for this descriptor -- the decision of gemm_launch itself, for callers that bill launches
// to roofline families (bench.py / scripts/pmc_summary.py through detr_tf/_hip.py; ADVICE r5: the Python copy of these rules had drifted).
// 0 = tile engine (4-wave), 1 = streaming kernel, 2 = ring kernel (bf16), 3 = ring kernel (fp32), 4 = GEMM + LayerNorm, 5 = ring weight gradient;
// negative: the descriptor is rejected (detr_hip_last_error).
extern "C" int detr_hip_gemm_family(const detr_gemm_desc *d) {
    GemmPlan p;
    if (gemm_prepare(d, p)) return -1;
    if (d->ln_y) return 4;
    RingPlan rp;
    const bool ring_first = tune(T_GEMM_STREAM) != 3 && gemm_ring_eligible(p, rp);
    if (!ring_first && gemm_stream_eligible(p)) return 1;
    if (ring_first) return 2;
    if (gemm_ring_f32_eligible(p, rp)) return 3;
    const GemmArgs &g = p.g;
    const bool ring_wgrad = p.bf16c && (p.tile == 1 || p.tile == 6 || p.tile == 7) && g.a16 && g.b16 && !p.ak && !p.bk && p.batch == 1 && p.split > 1 &&
                            g.slab_ts && !g.rowsum && d->M % 8 == 0 && d->N % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && aligned16(d->A) &&
                            aligned16(d->B) && tune(T_GEMM_RING) != 2 && tune(T_GEMM_RING) != 3;
    return ring_wgrad ? 5 : 0;
}

extern "C" int detr_hip_gemm_f32(const detr_gemm_desc *d, void *stream) {
    GemmPlan p;
    if (gemm_prepare(d, p)) return -1;
    return gemm_launch(p, (hipStream_t)stream);
}

// ---- grouped launch ------------------------------------------------------------------------------
template <bool AK, bool BKC>
static void launch_group_f32(const GemmGroupArgs &G, dim3 grid, hipStream_t s, bool split3) {
    if (split3) hipLaunchKernelGGL((gemm_x3_group_kernel<64, 64, AK, BKC>), grid, dim3(GEMM_THREADS), 0, s, G);
    else hipLaunchKernelGGL((gemm_f32_group_kernel<64, 64, 2, 2, AK, BKC>), grid, dim3(GEMM_THREADS), 0, s, G);
}
template <bool AK, bool BKC, bool A16, bool B16>
static void launch_group_bf16(const GemmGroupArgs &G, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((gemm_bf16c_group_kernel<64, 64, 2, 2, AK, BKC, A16, B16>), grid, dim3(GEMM_THREADS), 0, s, G);
}

extern "C" int detr_hip_gemm_group_f32(const detr_gemm_desc *descs, int32_t n, void *stream) {
    DETR_REQUIRE(descs != nullptr && n >= 1 && n <= 64, "gemm group: 1..64 descriptors");
    hipStream_t s = (hipStream_t)stream;
    int done = 0;
    while (done < n) {
        const int m = (n - done) < GEMM_MAX_GROUP ? (n - done) : GEMM_MAX_GROUP;
        GemmPlan p[GEMM_MAX_GROUP];
        for (int i = 0; i < m; ++i) {
            DETR_REQUIRE(!descs[done + i].ln_y, "gemm group: the fused LayerNorm (ln_y) exists for single launches only");
            if (gemm_prepare(descs + done + i, p[i])) return -1;
        }
        // one launch needs ONE kernel variant: 64x64 tiles, same layouts / storage types, no batch; members may differ in shape
        bool same = m > 1 && tune(T_GEMM_GROUP) != 2;
        for (int i = 0; i < m && same; ++i)
            same = p[i].tile == 0 && p[i].batch == 1 && p[i].bf16c == p[0].bf16c && p[i].split3 == p[0].split3 && p[i].ak == p[0].ak && p[i].bk == p[0].bk &&
                   p[i].g.a16 == p[0].g.a16 && p[i].g.b16 == p[0].g.b16 && (p[i].split == 1 || p[i].partial) &&
                   !gemm_stream_eligible(p[i]);
        if (same) { RingPlan rp; for (int i = 0; i < m && same; ++i) same = !gemm_ring_eligible(p[i], rp) && !gemm_ring_f32_eligible(p[i], rp); }
        if (!same) {
            for (int i = 0; i < m; ++i)
                if (gemm_launch(p[i], s)) return -1;
            done += m;
            continue;
        }
        GemmGroupArgs G;
        G.n = m;
        G.work_off[0] = 0;
        for (int i = 0; i < m; ++i) {
            G.g[i] = p[i].g;
            G.g[i].tiles_m = cdiv(p[i].g.M, 64);
            G.g[i].tiles_n = cdiv(p[i].g.N, 64);
            G.work_off[i + 1] = G.work_off[i] + G.g[i].tiles_m * G.g[i].tiles_n * p[i].split;
        }
        for (int i = m; i < GEMM_MAX_GROUP; ++i) { G.g[i] = G.g[0]; G.work_off[i + 1] = G.work_off[m]; }
        const dim3 grid((unsigned)G.work_off[m]);      // exactly one workgroup per (member, split, tile): balanced over the XCDs
        const bool ak = p[0].ak, bk = p[0].bk, a16 = p[0].g.a16 != 0, b16 = p[0].g.b16 != 0;
#define DETR_GROUP_BF16(A16_, B16_)                                          \
    do {                                                                     \
        if (ak && bk) launch_group_bf16<true, true, A16_, B16_>(G, grid, s);    \
        else if (ak) launch_group_bf16<true, false, A16_, B16_>(G, grid, s);    \
        else if (bk) launch_group_bf16<false, true, A16_, B16_>(G, grid, s);    \
        else launch_group_bf16<false, false, A16_, B16_>(G, grid, s);           \
    } while (0)
        if (p[0].bf16c) {
            if (a16 && b16) DETR_GROUP_BF16(true, true);
            else if (a16) DETR_GROUP_BF16(true, false);
            else if (b16) DETR_GROUP_BF16(false, true);
            else DETR_GROUP_BF16(false, false);
        } else if (ak && bk) launch_group_f32<true, true>(G, grid, s, p[0].split3);
        else if (ak) launch_group_f32<true, false>(G, grid, s, p[0].split3);
        else if (bk) launch_group_f32<false, true>(G, grid, s, p[0].split3);
        else launch_group_f32<false, false>(G, grid, s, p[0].split3);
#undef DETR_GROUP_BF16
        DETR_LAUNCH_CHECK("gemm group");
        // the members' split-K reductions, again as one launch (or handed back to the caller: detr_gemm_desc.defer_out)
        ReduceGroupArgs R;
        int nr = 0;
        unsigned rx = 1;
        bool all_small = true, any_small = false;
        for (int i = 0; i < m; ++i) {
            const detr_gemm_desc *d = p[i].d;
            if (d->defer_out) {
                detr_reduce_desc *o = d->defer_out;
                o->splits = 0;
                if (p[i].partial) {
                    o->ws = d->workspace; o->splits = p[i].split; o->part_stride = p[i].part; o->rows = d->M; o->cols = d->N;
                    o->C = d->C; o->ldc = d->ldc; o->alpha = p[i].final_e.alpha; o->scale = p[i].final_e.scale;
                    o->rs_ws = d->rowsum_a ? d->workspace + (long long)p[i].split * p[i].part : nullptr;
                    o->rs_out = d->rowsum_a; o->rs_alpha = d->rowsum_alpha;
                    o->ts_bm = p[i].ts_bm; o->ts_bn = p[i].ts_bn; o->ts_tiles_n = p[i].ts_tn;
                }
                continue;
            }
            if (!p[i].partial) continue;
            ReduceOne &r = R.r[nr];
            r.ws = d->workspace; r.splits = p[i].split; r.part_stride = p[i].part; r.rows = d->M; r.cols = d->N;
            r.C = d->C; r.ldc = d->ldc; r.alpha = p[i].final_e.alpha; r.scale = p[i].final_e.scale;
            r.rs_ws = d->rowsum_a ? d->workspace + (long long)p[i].split * p[i].part : nullptr;
            r.rs_out = d->rowsum_a; r.rs_alpha = d->rowsum_alpha;
            r.ts_bm = p[i].ts_bm; r.ts_bn = p[i].ts_bn; r.ts_tn = p[i].ts_tn;
            bool small;
            reduce_plan(r, small);
            all_small = all_small && small;
            any_small = any_small || small;
            rx = (unsigned)r.nblocks > rx ? (unsigned)r.nblocks : rx;
            ++nr;
        }
        if (nr > 0 && (all_small || !any_small)) {
            for (int i = nr; i < GEMM_MAX_GROUP; ++i) R.r[i] = R.r[0];
            if (all_small) hipLaunchKernelGGL(splitk_reduce_group_kernel<16>, dim3(rx, (unsigned)nr), dim3(256), 0, s, R);
            else hipLaunchKernelGGL(splitk_reduce_group_kernel<4>, dim3(rx, (unsigned)nr), dim3(256), 0, s, R);
            DETR_LAUNCH_CHECK("gemm group split-k reduce");
        } else if (nr > 0) {
            for (int i = 0; i < nr; ++i) {
                const ReduceOne &r = R.r[i];
                launch_splitk_reduce(r.ws, r.splits, r.part_stride, r.rows, r.cols, r.C, r.ldc, r.alpha, r.scale, s, r.rs_ws, r.rs_out,
                                     r.rs_alpha, r.ts_bm, r.ts_bn, r.ts_tn);
            }
            DETR_LAUNCH_CHECK("gemm group split-k reduce (sequential)");
        }
        done += m;
    }
    return 0;
}


// Deferred split-K reductions (detr_gemm_desc.defer_out): n slab sets reduced by ceil(n / 16) launches per size class instead of
// one launch each -- the ~90 reductions of a training step were ~9 us latency-bound launches (rocprofv3: 0.9 ms per step).
extern "C" int detr_hip_splitk_reduce_many(const detr_reduce_desc *descs, int32_t n, void *stream) {
    DETR_REQUIRE(descs != nullptr && n >= 1, "splitk_reduce_many: bad args");
    hipStream_t s = (hipStream_t)stream;
    for (int pass = 0; pass < 2; ++pass) {           // pass 0: small outputs (16 split groups per block), pass 1: large ones
        ReduceManyArgs R;
        int cnt = 0;
        unsigned rx = 1;
        auto flush = [&]() {
            if (cnt == 0) return;
            for (int i = cnt; i < REDUCE_MANY; ++i) R.r[i] = R.r[0];
            if (pass == 0) hipLaunchKernelGGL(splitk_reduce_many_kernel<16>, dim3(rx, (unsigned)cnt), dim3(256), 0, s, R);
            else hipLaunchKernelGGL(splitk_reduce_many_kernel<4>, dim3(rx, (unsigned)cnt), dim3(256), 0, s, R);
            cnt = 0;
            rx = 1;
        };
        for (int i = 0; i < n; ++i) {
            const detr_reduce_desc &d = descs[i];
            if (d.splits <= 0) continue;
            DETR_REQUIRE(d.ws && d.C && d.rows > 0 && d.cols > 0, "splitk_reduce_many: entry %d is malformed", i);
            ReduceOne r;
            r.ws = d.ws; r.splits = d.splits; r.part_stride = d.part_stride; r.rows = d.rows; r.cols = d.cols; r.C = d.C; r.ldc = d.ldc;
            r.alpha = d.alpha; r.scale = d.scale; r.rs_ws = d.rs_ws; r.rs_out = d.rs_out; r.rs_alpha = d.rs_alpha;
            r.ts_bm = d.ts_bm; r.ts_bn = d.ts_bn; r.ts_tn = d.ts_tiles_n > 0 ? d.ts_tiles_n : 1;
            DETR_REQUIRE(d.ts_bm == 0 || ((d.ts_bm == 64 || d.ts_bm == 128 || d.ts_bm == 256) && (d.ts_bn == 64 || d.ts_bn == 128 || d.ts_bn == 256) && d.ts_tiles_n > 0 &&
                                          d.part_stride % ((long long)d.ts_bm * d.ts_bn) == 0),
                         "splitk_reduce_many: entry %d has a malformed tile-ordered slab description", i);
            bool small;
            reduce_plan(r, small);
            if (small != (pass == 0)) continue;
            R.r[cnt++] = r;
            rx = (unsigned)r.nblocks > rx ? (unsigned)r.nblocks : rx;
            if (cnt == REDUCE_MANY) flush();
        }
        flush();
    }
    DETR_LAUNCH_CHECK("splitk_reduce_many");
    return 0;
}
