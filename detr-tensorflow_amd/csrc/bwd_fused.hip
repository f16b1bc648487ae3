// bwd_fused.hip -- backward of a bottleneck 1x1 convolution 64 -> 256 channels that reads dY ONCE (round 5).
// Reference: resnet_backbone.py:116-137 (conv3 / the projection shortcut of a BottleNeck), optimizers.py:110-120 (their kernels train).
//
// The backward of y = conv1x1(a) (a: [M][64], y: [M][256], W: [64][256]) needs
//     da[M][64]   = mask(a > 0) . (dY W^T)            (input gradient, the ReLU of the layer below folded in)
//     dW[64][256] = a^T dY                            (weight gradient)
// and both read dY -- 274 MB at M = 534 400 (layer1 of the B = 8, 800 x 1333 step).  As two launches (streaming GEMM 84 us + split-K GEMM 63 us,
// scripts/experiments/dy_once_probe.py) they move 410 + 342 MB; chunking the pair so that the second read of a chunk could come out of the
// 256 MB memory-side cache does not help (same probe: 2 / 4 / 8 chunks 156 / 188 / 257 us against 147).  This kernel streams dY and `a` once:
//   * one persistent workgroup per CU walks a contiguous range of 32-row strips.  A strip -- dY rows (16 KB) and `a` rows (4 KB) -- is
//     requested by LDS-DMA three strips ahead into a ring of four 20 KB stages (counted vmcnt, one raw s_barrier per strip; the pieces are
//     inline assembly for the reason given in gemm_ring.h); W (32 KB) is resident in LDS.
//   * six waves with fixed roles: waves 0-3 issue the requests (5 pieces each per strip) and each accumulates dW[:, 64 w .. 64 w + 63]
//     (8 MFMAs per strip: both operands are k-major = row-major strips, read with ds_read_b64_tr_b16); waves 4-5 compute the input gradient
//     of the strip, 32 output columns each (16 MFMAs, operands swapped so that a lane owns one row: W fragments and dY row fragments are
//     plain ds_read_b128), apply the mask out of the `a` strip that is already in LDS, and hand the packed bf16 rows to each other through
//     a small staging area so that the stores of a strip -- issued one strip later -- cover whole 128-byte lines.
//   * dY strip image: the row-major transpose-read image of gemm_ring.h (pieces of 4 rows x 128 columns, 16-byte chunk pc of row kr at
//     position pc ^ 4 kr) with one more XOR term, the row group's upper bits (rg >> 1): it permutes the chunks of all four rows of a group
//     alike, so the transpose reads stay conflict-free, and makes the row fragments of the input gradient (32 rows, one chunk column)
//     conflict-free as well (the 16 rows of a ds_read_b128 service group have distinct (row & 3, row >> 3) pairs).
//     `a` strip image: 128-byte rows, chunk c of row r at position c ^ 4 ((r >> 1) & 1) (rows r and r + 2 would share banks otherwise).
//   * every workgroup writes its dW partial as one fp32 slab; the ordered split-K reduction of gemm_f32.hip sums the slabs (deterministic),
//     applies scale[n] (the folded BatchNorm) and accumulates into the gradient.
// The input gradient accumulates k ascending in one fp32 accumulator per output, like the streaming kernel: bit-identical to it.
// HBM per launch: 274 + 68 + 68 MB + 2 x 17 MB of slabs instead of 752 MB.
#include "common.h"
#include "gemm_ring.h"

namespace detr {

#ifndef DETR_BF_NS
#define DETR_BF_NS 4
#endif
#ifndef DETR_BF_INTERLEAVE
#define DETR_BF_INTERLEAVE 0   // 1: workgroup w takes strips w, w + nwg, w + 2 nwg, ... (one sweep over the tensors) instead of a contiguous range (A/B builds)
#endif
#ifndef DETR_BF_NT
#define DETR_BF_NT 0           // 1: the strip requests carry the non-temporal hint (A/B builds)
#endif
constexpr int BF_D1 = 64, BF_D2 = 256, BF_SR = 32, BF_NS = DETR_BF_NS;
constexpr int BF_WAVES = 6, BF_THREADS = 64 * BF_WAVES;
constexpr int BF_W_BYTES = BF_D1 * BF_D2 * 2;                    // 32 KB
constexpr int BF_G_BYTES = BF_SR * BF_D2 * 2, BF_Y_BYTES = BF_SR * BF_D1 * 2, BF_STAGE = BF_G_BYTES + BF_Y_BYTES;      // 16 + 4 KB
constexpr int BF_OFF_RING = BF_W_BYTES;
constexpr int BF_ST_PITCH = 144;                                  // bytes per staged output row (128 + 16)
constexpr int BF_OFF_ST = BF_OFF_RING + BF_NS * BF_STAGE;
constexpr int BF_SMEM = BF_OFF_ST + 2 * BF_SR * BF_ST_PITCH;      // 123 904 bytes: one workgroup per CU
constexpr int BF_PW = 5;                                          // DMA pieces per requesting wave and strip

struct BwdFusedArgs {
    const unsigned short *g; long long ldg;       // dY [M][ldg], 256 columns
    const unsigned short *y; long long ldy;       // a  [M][ldy], 64 columns: weight-gradient operand and (use_mask) the ReLU mask
    const unsigned short *w; long long ldw;       // W  [64][ldw], 256 columns contiguous
    unsigned short *dz; long long lddz;           // da [M][lddz], 64 columns
    float *slabs;                                 // [nwg][64][256]
    int M, nstrips, nwg, use_mask;
};

__global__ __launch_bounds__(BF_THREADS, 1) void conv1x1_bwd_fused_bf16_kernel(BwdFusedArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(1024))) char bf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = ring_lds_addr(bf_smem);
    const int wg = blockIdx.x;
    // strips of this workgroup: v = 0 .. cnt - 1 -> strip sv(v)
    const int s0 = DETR_BF_INTERLEAVE ? wg : (int)((long long)a.nstrips * wg / a.nwg);
    const int cnt = DETR_BF_INTERLEAVE ? (a.nstrips - wg + a.nwg - 1) / a.nwg : (int)((long long)a.nstrips * (wg + 1) / a.nwg) - s0;
    const int sstep = DETR_BF_INTERLEAVE ? a.nwg : 1;
    auto sv = [&](const int v) { return s0 + v * sstep; };
    const int l31 = lane & 31, hh = lane >> 5;

    if (wave < 4) {
        // ================= requests + weight gradient of columns [64 wave, 64 wave + 64) =================
        // dY pieces P = wave + 4 i (i < 4): rows 4 (P >> 1) .. + 3, column half P & 1; lane: row kr = lane >> 4, slot position lane & 15
        unsigned gvoff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int P = wave + 4 * i, kr = lane >> 4, ps = lane & 15;
            const int pc = ps ^ (4 * kr) ^ (P >> 2);
            gvoff[i] = (unsigned)(4 * (P >> 1) + kr) * (unsigned)a.ldg * 2u + (unsigned)((P & 1) * 256 + pc * 16);
        }
        // `a` piece = rows 8 wave .. + 7; lane: row 8 wave + lane / 8, slot position lane & 7
        unsigned yvoff;
        {
            const int r = 8 * wave + (lane >> 3), pos = lane & 7;
            yvoff = (unsigned)r * (unsigned)a.ldy * 2u + 16u * (unsigned)(pos ^ (4 * ((r >> 1) & 1)));
        }
        auto issue_strip = [&](const int v, const int stage) {
            const long long row0 = (long long)sv(v) * BF_SR;
            const long long left = (v < cnt && row0 < a.M) ? (long long)a.M - row0 : 0;            // rows of the strip onwards (0: nothing to fetch)
            const unsigned long long gb = left > 0 ? (unsigned long long)((left - 1) * a.ldg + BF_D2) * 2ull : 0ull;
            const unsigned long long yb = left > 0 ? (unsigned long long)((left - 1) * a.ldy + BF_D1) * 2ull : 0ull;
            const u32x4 rg = ring_rsrc(reinterpret_cast<const char *>(a.g) + row0 * a.ldg * 2, (unsigned)(gb > 0xFFFFFFFFull ? 0xFFFFFFFFull : gb));
            const u32x4 ry = ring_rsrc(reinterpret_cast<const char *>(a.y) + row0 * a.ldy * 2, (unsigned)(yb > 0xFFFFFFFFull ? 0xFFFFFFFFull : yb));
            const unsigned base = lds0 + (unsigned)(BF_OFF_RING + stage * BF_STAGE);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (DETR_BF_NT != 0) ring_dma_piece_nt(rg, base + (unsigned)((wave + 4 * i) * 1024), gvoff[i]);
                else ring_dma_piece(rg, base + (unsigned)((wave + 4 * i) * 1024), gvoff[i]);
            }
            if constexpr (DETR_BF_NT != 0) ring_dma_piece_nt(ry, base + (unsigned)(BF_G_BYTES + wave * 1024), yvoff);
            else ring_dma_piece(ry, base + (unsigned)(BF_G_BYTES + wave * 1024), yvoff);
        };
        // W image: row n = 512 bytes, chunk c at position c ^ (n & 15); piece = 2 rows, lane: row 2 P + lane / 32, slot position lane & 31
        {
            const u32x4 rw = ring_rsrc(a.w, (unsigned)(((long long)(BF_D1 - 1) * a.ldw + BF_D2) * 2));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int P = wave + 4 * i, n = 2 * P + (lane >> 5), pos = lane & 31;
                ring_dma_piece(rw, lds0 + (unsigned)(P * 1024), (unsigned)n * (unsigned)a.ldw * 2u + 16u * (unsigned)(pos ^ (n & 15)));
            }
        }
#pragma unroll
        for (int t = 0; t < BF_NS - 1; ++t) issue_strip(t, t);

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        // transpose-read lane offsets: lane = 16 gq + t hands in the 8-byte chunk (row kr = t >> 2, quarter q = t & 3) of sub-block (gq & 1)
        const int gq = lane >> 4, tq = lane & 15, kr = tq >> 2, q = tq & 3;
        int goff[2], yoff[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {      // dY columns 64 wave + 32 nb ..: 16-column sub-block ibg = (64 wave + 32 nb) / 16 + (gq & 1); row group 4 kk + 2 (gq >> 1) (+ 1)
            const int ibg = 4 * wave + 2 * nb + (gq & 1), nh = ibg >> 3, ib = ibg & 7;
            goff[nb] = (2 * (gq >> 1) * 2 + nh) * 1024 + kr * 256 + ((((2 * ib + (q >> 1)) ^ (4 * kr)) ^ (gq >> 1)) * 16) + (q & 1) * 8;
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)        // `a` columns 32 mb ..: rows 16 kk + 8 (gq >> 1) + 4 half + kr
            yoff[mb] = BF_G_BYTES + (gq >> 1) * 1024 + kr * 128 + (((4 * mb + 2 * (gq & 1) + (q >> 1)) ^ (4 * (kr >> 1))) * 16) + (q & 1) * 8;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        auto tr2 = [&](const char *p, int second) -> bf16x8 {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + second));
            const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(bf16x8, v);
        };
        int stage = 0, nxt = BF_NS - 1;                  // ring slots of strip s and of strip s + NS - 1
        for (int v = 0; v < cnt; ++v) {
            ring_wait_vmcnt<(BF_NS - 2) * BF_PW>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_strip(v + BF_NS - 1, nxt);
            const char *st = bf_smem + BF_OFF_RING + stage * BF_STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 fy[2], fg[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) fy[mb] = tr2(st + yoff[mb] + kk * 2048, 512);                      // rows + 4: four 128-byte rows on
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) fg[nb] = tr2(st + ((goff[nb] ^ (32 * kk)) + kk * 8192), 2048);     // row group + 1: two pieces on
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[mb], fg[nb], acc[mb][nb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            stage = (stage + 1 == BF_NS) ? 0 : stage + 1;
            nxt = (nxt + 1 == BF_NS) ? 0 : nxt + 1;
        }
        ring_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                    // (pairs with the other role's last barrier)
        // slab: dW[m][n], lane holds column n = l31 of rows (r & 3) + 8 (r >> 2) + 4 hh of each 32 x 32 block
        float *slab = a.slabs + (long long)wg * (BF_D1 * BF_D2);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[(32 * mb + (r & 3) + 8 * (r >> 2) + 4 * hh) * BF_D2 + 64 * wave + 32 * nb + l31] = acc[mb][nb][r];
    } else {
        // ================= input gradient: wave 4 + d owns output columns [32 d, 32 d + 32) of every strip =================
        const int d = wave - 4;
        const int n = 32 * d + l31;
        const int woff = n * 512;                                   // W fragment of k-step kk: chunk (2 kk + hh) ^ (n & 15)
        const int wsw = n & 15;
        const int grow = (l31 >> 2) * 2048 + (l31 & 3) * 256;       // dY row fragment: piece pair of row group l31 >> 2, row l31 & 3
        const int gsw = (4 * (l31 & 3)) ^ (l31 >> 3);
        unsigned short *dz = a.dz;
        int stage = 0;
        auto store_strip = [&](const int v) {        // rows 16 d .. 16 d + 15 of this workgroup's strip v out of staging buffer v & 1: two passes of 8 rows x 128 bytes
            const int s = sv(v);
            const char *sb = bf_smem + BF_OFF_ST + (v & 1) * (BF_SR * BF_ST_PITCH);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = 16 * d + 8 * i + (lane >> 3), c = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4 *>(sb + r * BF_ST_PITCH + c * 16);
                const long long row = (long long)s * BF_SR + r;
                if (row < a.M) *reinterpret_cast<uint4 *>(dz + row * a.lddz + c * 8) = v;
            }
        };
        for (int v = 0; v < cnt; ++v) {
            // (a raw s_barrier orders nothing by itself: the staged rows of the previous strip must have been WRITTEN before the other wave reads them)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (v > 0) store_strip(v - 1);
            const char *st = bf_smem + BF_OFF_RING + stage * BF_STAGE;
            stage = (stage + 1 == BF_NS) ? 0 : stage + 1;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int c32 = 2 * kk + hh;                         // 16-byte chunk of the 512-byte row
                const bf16x8 fw = *reinterpret_cast<const bf16x8 *>(bf_smem + woff + ((c32 ^ wsw) * 16));
                const bf16x8 fg = *reinterpret_cast<const bf16x8 *>(st + grow + (kk >> 3) * 1024 + ((((c32 & 15) ^ gsw)) * 16));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fg, acc, 0, 0, 0);
            }
            // lane: row l31, columns 32 d + 8 j + 4 hh .. + 3 in acc[4 j .. 4 j + 3]
            char *sb = bf_smem + BF_OFF_ST + (v & 1) * (BF_SR * BF_ST_PITCH);
            const char *ym = st + BF_G_BYTES + (l31 >> 3) * 1024 + (l31 & 7) * 128 + hh * 8;
            const int ysw = 4 * ((l31 >> 1) & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
                if (a.use_mask) {
                    const uint2 m = *reinterpret_cast<const uint2 *>(ym + (((4 * d + j) ^ ysw) * 16));
                    const unsigned mw[2] = {m.x, m.y};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float mv = bf16_bits_to_f32((e & 1) ? (mw[e >> 1] >> 16) : (mw[e >> 1] & 0xFFFFu));
                        v[e] = (mv > 0.0f) ? v[e] : 0.0f;
                    }
                }
                *reinterpret_cast<uint2 *>(sb + l31 * BF_ST_PITCH + (32 * d + 8 * j + 4 * hh) * 2) = make_uint2(f32_to_bf16_pair(v[0], v[1]), f32_to_bf16_pair(v[2], v[3]));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (cnt > 0) store_strip(cnt - 1);
    }
#endif
}

}  // namespace detr

using namespace detr;

// Workspace floats the fused backward needs for (M rows): one 64 x 256 slab per workgroup
extern "C" int64_t detr_hip_conv1x1_bwd_fused_workspace_floats(int64_t M) {
    const int64_t nstrips = (M + BF_SR - 1) / BF_SR;
    const int64_t nwg = nstrips < 256 ? nstrips : 256;
    return nwg * BF_D1 * BF_D2;
}

// da = mask(a > 0) . (dY W^T) and dW += alpha * scale[n] * (a^T dY) in one pass over dY (see the header of this file).
// dY [M][ldg] (256 columns), a [M][lda] (64 columns), W [64][ldw], da [M][ldda] -- bf16; dW [64][lddw], scale [256] or NULL -- fp32.
extern "C" int detr_hip_conv1x1_bwd_fused_bf16(const uint16_t *dy, int64_t ldg, const uint16_t *a_in, int64_t lda, const uint16_t *w, int64_t ldw,
                                               uint16_t *da, int64_t ldda, int32_t use_mask, float *dw, int64_t lddw, const float *scale, float alpha,
                                               int64_t M, int32_t d1, int32_t d2, float *workspace, int64_t workspace_floats, void *stream) {
    DETR_REQUIRE(dy && a_in && w && da && dw && workspace && M > 0, "conv1x1 bwd (fused): bad operands");
    DETR_REQUIRE(d1 == BF_D1 && d2 == BF_D2, "conv1x1 bwd (fused): built for %d -> %d channels, got %d -> %d", BF_D1, BF_D2, d1, d2);
    DETR_REQUIRE(ldg >= BF_D2 && lda >= BF_D1 && ldw >= BF_D2 && ldda >= BF_D1 && lddw >= BF_D2 && ldg % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldda % 8 == 0,
                 "conv1x1 bwd (fused): leading dimensions");
    DETR_REQUIRE(aligned16(dy) && aligned16(a_in) && aligned16(w) && aligned16(da) && aligned16(dw) && aligned16(workspace), "conv1x1 bwd (fused): alignment");
    DETR_REQUIRE(M * ldg * 2 < (1ll << 32) && M * ldda * 2 < (1ll << 32), "conv1x1 bwd (fused): operand larger than a buffer descriptor's range");
    BwdFusedArgs k;
    k.g = dy; k.ldg = ldg; k.y = a_in; k.ldy = lda; k.w = w; k.ldw = ldw; k.dz = da; k.lddz = ldda;
    k.M = (int)M; k.nstrips = (int)((M + BF_SR - 1) / BF_SR); k.nwg = k.nstrips < 256 ? k.nstrips : 256; k.use_mask = use_mask;
    DETR_REQUIRE(workspace_floats >= (int64_t)k.nwg * BF_D1 * BF_D2, "conv1x1 bwd (fused): workspace holds %lld floats, %lld needed", (long long)workspace_floats,
                 (long long)k.nwg * BF_D1 * BF_D2);
    k.slabs = workspace;
    static bool reserved = false;
    if (!reserved) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_bwd_fused_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BF_SMEM);
        DETR_REQUIRE(e == hipSuccess, "conv1x1 bwd (fused): cannot reserve %d bytes of LDS: %s", BF_SMEM, hipGetErrorString(e));
        reserved = true;
    }
    hipLaunchKernelGGL(conv1x1_bwd_fused_bf16_kernel, dim3((unsigned)k.nwg), dim3(BF_THREADS), BF_SMEM, (hipStream_t)stream, k);
    DETR_LAUNCH_CHECK("conv1x1 bwd (fused)");
    launch_splitk_reduce(workspace, k.nwg, (long long)BF_D1 * BF_D2, BF_D1, BF_D2, dw, lddw, alpha, scale, (hipStream_t)stream, nullptr, nullptr, 0.0f, 0, 0, 0);
    DETR_LAUNCH_CHECK("conv1x1 bwd (fused): reduction");
    return 0;
}
