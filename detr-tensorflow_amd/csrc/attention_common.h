// attention_common.h -- definitions shared by the exact-fp32 (attention_f32.hip) and the bf16-MFMA
// (attention_bf16.hip) fused attention kernels.
#pragma once
#include "common.h"

namespace detr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AT_KEYS = 32;      // keys (or queries, in the dK/dV kernel) per LDS tile

struct AttnArgs {
    const float *Q, *K, *V;      // [B, T|S, ld]
    float *O;                    // fwd out / bwd in
    float *LSE;                  // [B*H, T] log-sum-exp of every score row
    const float *dO;
    float *dQ, *dK, *dV;
    float *delta;                // [B*H, T] rowsum(dO * O)
    int B, H, T, S;
    // row strides (floats) of the batch-first token matrices: every tensor has its own, so that Q / K / V (and their
    // gradients) may be column blocks of one packed projection buffer ([rows, 768] self-attention, [rows, 12*256] for the
    // layer-invariant decoder cross-attention K / V)
    long long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    float qscale;           // softmax(qscale * q.k): folded into the log2(e) factor Q is loaded with; dQ is returned w.r.t. the UNSCALED q
    float drop_scale;       // 1/(1-p) or 0
    uint32_t drop_thresh, drop_seed;      // drop_seed = dropout SITE id
    const uint32_t *drop_step;            // per-step seed in device memory (may be null), see common.h drop_key
};

__device__ __forceinline__ int krow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

constexpr float AT_LOG2E = 1.4426950408889634f;
constexpr float AT_LN2 = 0.6931471805599453f;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32; exp2(-inf) = 0

// Keep flags of the 16 scores a lane holds (keys kbase + krow(r, hi), r = 0..15): adjacent keys (r, r + 1), r even,
// share one hash (common.h drop_hash).  rowbase = row * Sp and kbase are even.  f(r, keep) is called for r = 0..15.
// Same function as drop_keep(), restated so that a tile costs 8 x (1 add + 5 xor/shift + 2 multiplies) + 16 compares:
//   * the 64-bit pair index is formed ONCE per tile (rowbase + kbase + 4 hi) / 2; the eight pairs of a lane are small
//     constants above it.  Its high word only enters the hash as hi * 0x9E3779B9, folded into the key once per tile;
//     a low word within 32 of wrapping (never for < 2^33 elements) takes the plain drop_hash path;
//   * the last round x ^= x >> 16 only changes the LOW half: the high element compares the whole word with thresh << 16;
//   * the flags go straight to the caller (v_cmp -> v_cndmask), not through a bit mask.
template <typename F>
__device__ __forceinline__ void drop_keep16(uint32_t key, unsigned long long rowbase, int kbase, int hi, uint32_t thresh16, F &&f) {
    const unsigned long long tb = (rowbase + (unsigned)(kbase + 4 * hi)) >> 1;
    const uint32_t tlo = (uint32_t)tb, thi = (uint32_t)(tb >> 32);
    if (tlo > 0xFFFFFFDFu) {                     // (tlo + 13 could carry into the high word: generic form)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const uint32_t h = drop_hash(key, tb + (unsigned)(((r & 3) + 8 * (r >> 2)) >> 1));
            f(r, (h & 0xFFFFu) >= thresh16);
            f(r + 1, (h >> 16) >= thresh16);
        }
        return;
    }
    const uint32_t key2 = key * 0x85EBCA6Bu + 0xC2B2AE35u;
    const uint32_t kx = key ^ (thi * 0x9E3779B9u);
    const uint32_t th_hi = thresh16 << 16;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        uint32_t x = (tlo + (uint32_t)(((r & 3) + 8 * (r >> 2)) >> 1)) ^ kx;
        x ^= x >> 16; x *= 0x7feb352du;
        x ^= key2;
        x ^= x >> 15; x *= 0x846ca68bu;
        f(r, ((x ^ (x >> 16)) & 0xFFFFu) >= thresh16);
        f(r + 1, x >= th_hi);
    }
}

// The same flags for the dK / dV kernels, where a lane holds ONE key (sk, parity = lane & 1) and 16 query rows
// qbase + krow(r, hi): f(r, keep) for r = 0..15.  One 32-bit hash serves the key PAIR (sk & ~1, sk | 1) of a query row, and
// the partner key lives in lane ^ 1: each lane hashes 8 of its 16 rows (registers r8 + 8 * parity) and takes the other 8
// words from its neighbour with one DPP quad_perm each.  The 64-bit pair index (row * Sp + sk) / 2 = row * Sp/2 + sk/2 is formed
// once per tile; the eight rows of a lane are multiples of Sp / 2 above it (a low word that could carry: plain drop_hash).
// This lane's key uses the high half of a word when sk is odd, the low half otherwise: the half is shifted to the top and
// the whole word compared with thresh << 16.
template <typename F>
__device__ __forceinline__ void drop_keep16_keycol(uint32_t key, unsigned long long row0, unsigned long long Sp, int sk, int lane, int hi,
                                                   uint32_t thresh16, F &&f) {
    const unsigned halfSp = (unsigned)(Sp >> 1);
    const int odd = lane & 1;                            // == sk & 1 (a workgroup's first key is even)
    const unsigned long long pb = row0 * halfSp + (unsigned)(sk >> 1) + (unsigned long long)(unsigned)(4 * hi + 16 * odd) * halfSp;
    const uint32_t tlo = (uint32_t)pb, thi = (uint32_t)(pb >> 32);
    const uint32_t key2 = key * 0x85EBCA6Bu + 0xC2B2AE35u;
    const uint32_t kx = key ^ (thi * 0x9E3779B9u);
    const bool slow = tlo > 0xFFFFFFFFu - 12u * halfSp;
    uint32_t hown[8], hoth[8];
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) {
        const unsigned c = (unsigned)((r8 & 3) + 8 * (r8 >> 2));         // krow(r8, hi) - 4 hi
        uint32_t x = (tlo + c * halfSp) ^ kx;
        x ^= x >> 16; x *= 0x7feb352du;
        x ^= key2;
        x ^= x >> 15; x *= 0x846ca68bu;
        x ^= x >> 16;
        hown[r8] = slow ? drop_hash(key, pb + (unsigned long long)c * halfSp) : x;
    }
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8)
        hoth[r8] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hown[r8], 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    const uint32_t sh = odd ? 0u : 16u, th_hi = thresh16 << 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t hw = ((r >> 3) == odd) ? hown[r & 7] : hoth[r & 7];
        f(r, (hw << sh) >= th_hi);
    }
}

// maximum / sum of the 16 registers of a score column as trees (depth 4 instead of a 15-deep dependent chain: with ~2
// waves per SIMD the chain latency, not the issue rate, is what a tile waits for)
__device__ __forceinline__ float tree_max16(const f32x16 &s) {
    const float a0 = fmaxf(fmaxf(s[0], s[1]), s[2]), a1 = fmaxf(fmaxf(s[3], s[4]), s[5]), a2 = fmaxf(fmaxf(s[6], s[7]), s[8]);
    const float a3 = fmaxf(fmaxf(s[9], s[10]), s[11]), a4 = fmaxf(fmaxf(s[12], s[13]), s[14]);
    return fmaxf(fmaxf(fmaxf(a0, a1), a2), fmaxf(fmaxf(a3, a4), s[15]));
}
__device__ __forceinline__ float tree_sum16(const float (&p)[16]) {
    const float a0 = (p[0] + p[1]) + (p[2] + p[3]), a1 = (p[4] + p[5]) + (p[6] + p[7]);
    const float a2 = (p[8] + p[9]) + (p[10] + p[11]), a3 = (p[12] + p[13]) + (p[14] + p[15]);
    return (a0 + a1) + (a2 + a3);
}
// the two halves of a wave hold the two halves of every score column (hi = lane >> 5): combine across them with
// v_permlane32_swap (one VALU instruction; __shfl_xor(x, 32) is a ds_bpermute round trip through the LDS pipe).
// After the swap a = [x.lo | x.lo], b = [x.hi | x.hi]: every lane sees both halves of its column.
__device__ __forceinline__ void half_swap(float &a, float &b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float halves_max(float x) { float a = x, b = x; half_swap(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float halves_sum(float x) { float a = x, b = x; half_swap(a, b); return a + b; }


static int attn_check_ld(long long ld, int H, const char *what) {
    DETR_REQUIRE(ld >= (long long)H * 32 && ld % 4 == 0, "attention: row stride of %s (%lld) must be >= heads*32 and a multiple of 4", what, ld);
    return 0;
}

// fills AttnArgs from the C-ABI descriptor; bwd = 1 also checks the gradient operands
static int attn_from_desc(const detr_attn_desc *d, int bwd, AttnArgs &a) {
    DETR_REQUIRE(d, "attention: null descriptor");
    DETR_REQUIRE(d->q && d->k && d->v && d->o && d->lse, "attention: null operand");
    DETR_REQUIRE(d->B > 0 && d->H > 0 && d->T > 0 && d->S > 0, "attention: bad shape B=%d H=%d T=%d S=%d", d->B, d->H, d->T, d->S);
    if (attn_check_ld(d->ldq, d->H, "q") || attn_check_ld(d->ldk, d->H, "k") || attn_check_ld(d->ldv, d->H, "v") ||
        attn_check_ld(d->ldo, d->H, "o")) return -1;
    DETR_REQUIRE(aligned16(d->q) && aligned16(d->k) && aligned16(d->v) && aligned16(d->o), "attention: operands must be 16-byte aligned");
    DETR_REQUIRE((long long)d->B * d->H <= 65535, "attention: B*H=%lld exceeds grid.y", (long long)d->B * d->H);
    DETR_REQUIRE(d->dropout_p >= 0.0f && d->dropout_p < 1.0f, "attention: dropout p=%f out of range", d->dropout_p);
    DETR_REQUIRE(d->scale > 0.0f, "attention: scale must be positive (1 = q already scaled)");
    a = AttnArgs{};
    a.Q = d->q; a.K = d->k; a.V = d->v; a.O = d->o; a.LSE = d->lse;
    a.B = d->B; a.H = d->H; a.T = d->T; a.S = d->S;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
    a.qscale = d->scale;
    a.drop_scale = d->dropout_p > 0.0f ? 1.0f / (1.0f - d->dropout_p) : 0.0f;
    a.drop_thresh = drop_thresh16(d->dropout_p);
    a.drop_seed = d->dropout_site;
    a.drop_step = d->dropout_step;
    if (bwd) {
        DETR_REQUIRE(d->d_o && d->dq && d->dk && d->dv && d->delta, "attention bwd: null operand");
        if (attn_check_ld(d->ldd_o, d->H, "d_o") || attn_check_ld(d->lddq, d->H, "dq") || attn_check_ld(d->lddk, d->H, "dk") ||
            attn_check_ld(d->lddv, d->H, "dv")) return -1;
        DETR_REQUIRE(aligned16(d->d_o) && aligned16(d->dq) && aligned16(d->dk) && aligned16(d->dv), "attention bwd: gradients must be 16-byte aligned");
        a.dO = d->d_o; a.dQ = d->dq; a.dK = d->dk; a.dV = d->dv; a.delta = d->delta;
        a.lddo = d->ldd_o; a.lddq = d->lddq; a.lddk = d->lddk; a.lddv = d->lddv;
    }
    return 0;
}

// waves per workgroup: 4 waves (128 rows) amortise the K/V tile loads best, but the grid must still balance over
// 256 CUs -- below ~8 workgroups per CU the 2-wave kernels (twice the workgroups) win.  DETR_HIP_ATTN_WAVES forces.
static int attn_waves(int rows, int bh) {
    const int force = tune(T_ATTN_WAVES);
    if (force == 2 || force == 4) return force;
    return ((long long)cdiv(rows, 128) * bh < 2048) ? 2 : 4;
}

int attn_fwd_bf16_launch(const AttnArgs &a, hipStream_t s);      // attention_bf16.hip
int attn_bwd_bf16_launch(const AttnArgs &a, hipStream_t s);
int attn2_fwd_from_desc(const detr_attn_desc *d, hipStream_t s);    // attention_dma.hip (io_dtype = 1)
int attn2_bwd_from_desc(const detr_attn_desc *d, hipStream_t s);

}  // namespace detr
