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
    long long ld;
    float drop_scale;       // 1/(1-p) or 0
    uint32_t drop_thresh, drop_seed;
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


static int attn_check_args(const float *q, const float *k, const float *v, int B, int H, int T, int S, long long ld) {
    DETR_REQUIRE(q && k && v, "attention: null operand");
    DETR_REQUIRE(B > 0 && H > 0 && T > 0 && S > 0, "attention: bad shape B=%d H=%d T=%d S=%d", B, H, T, S);
    DETR_REQUIRE(ld >= (long long)H * 32 && ld % 4 == 0, "attention: row stride %lld must be >= heads*32 and a multiple of 4", ld);
    DETR_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v), "attention: operands must be 16-byte aligned");
    DETR_REQUIRE((long long)B * H <= 65535, "attention: B*H=%lld exceeds grid.y", (long long)B * H);
    return 0;
}

// waves per workgroup: 4 waves (128 rows) amortise the K/V tile loads best, but the grid must still balance over
// 256 CUs -- below ~8 workgroups per CU the 2-wave kernels (twice the workgroups) win.  DETR_HIP_ATTN_WAVES forces.
static int attn_waves(int rows, int bh) {
    const int force = env_tile("DETR_HIP_ATTN_WAVES");
    if (force == 2 || force == 4) return force;
    return ((long long)cdiv(rows, 128) * bh < 2048) ? 2 : 4;
}

static int attn_set_drop(AttnArgs &a, float p, uint32_t seed) {
    DETR_REQUIRE(p >= 0.0f && p < 1.0f, "attention: dropout p=%f out of range", p);
    a.drop_scale = p > 0.0f ? 1.0f / (1.0f - p) : 0.0f;
    a.drop_thresh = drop_thresh16(p);
    a.drop_seed = seed;
    return 0;
}

}  // namespace detr
