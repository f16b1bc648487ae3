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

// keep flags of the 16 scores a lane holds (keys kbase + krow(r, hi), r = 0..15): adjacent keys (r, r+1), r even,
// share one hash.  rowbase = row * Sp (even), bit r of the result = keep.
__device__ __forceinline__ uint32_t keep_bits16(uint32_t seed, unsigned long long rowbase, int kbase, int hi, uint32_t thresh16) {
    uint32_t bits = 0;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const uint32_t h = drop_hash(seed, (rowbase + (unsigned)(kbase + krow(r, hi))) >> 1);
        bits |= ((h & 0xFFFFu) >= thresh16 ? 1u : 0u) << r;
        bits |= ((h >> 16) >= thresh16 ? 1u : 0u) << (r + 1);
    }
    return bits;
}


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

}  // namespace detr
