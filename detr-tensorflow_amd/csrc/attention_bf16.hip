// attention_bf16.hip -- the fused attention core of attention_f32.hip with bf16 MFMA operands (precision="bf16",
// BASELINE config C3): Q, K, V, P, dO and dS are rounded to bf16 (RNE) on their way into registers / LDS, the products
// run on v_mfma_f32_32x32x16_bf16 (2 instructions per 32x32x32 product instead of 16 v_mfma_f32_32x32x2_f32), the
// accumulation, the softmax statistics, LSE / delta and all outputs stay fp32.  Same transposed formulation:
//     S^T[key][query] = K_tile (A: 8 consecutive d of a key row, ds_read_b128) x Q^T (B: per-lane registers)
//   C/D map: lane l holds query (l&31) and the 16 keys krow(r, hi), hi = l>>5.
//     O^T[d][query] += V^T (A) x P^T (B = the registers above, packed to bf16):  the k slot j (0..7) of k-step s2 of
//   lane half hi is DEFINED as key 16*s2 + 4*hi + (j&3) + 8*(j>>2) -- exactly the keys the lane already holds in
//   registers 8*s2 .. 8*s2+7 -- and the A operand follows: V stays [key][d] in LDS and ds_read_b64_tr_b16 (gfx950
//   transpose read; the 16 lanes of a group hand in the 8-byte chunks of four arbitrary rows and lane c receives
//   column c) gathers V[those four keys][d = lane's column].  No data movement between QK^T and PV, no V transpose.
// The backward kernels reuse the same two fragment forms (row fragments for the score-type products, transpose-read
// fragments for the gradient-type products).  Softmax arithmetic, masks and dropout as in attention_f32.hip.
#include "attention_common.h"
#include "gemm_core.h"        // BufSrc: buffer-descriptor loads

namespace detr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int AB_LD = 40;        // bf16 per LDS tile row (80 B): 16-byte aligned rows, conflict-free ds_read_b128 fragments
// (One 32-key tile per loop iteration: staging 2 / 4 sub-tiles per iteration and walking them in sequence measured 35 / 45 / 81 ->
//  37 / 45 / 76 -> 39 / 47 / 98 us (fwd / dQ / dKdV averages, round 1): sequential sub-tiles add no parallelism.  What does is
//  more WAVES per problem -- the in-workgroup split of the streamed dimension below.)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    bf16x2 r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ bf16x8 pack8(const float *v) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w;
    w[0] = pk_bf16(v[0], v[1]); w[1] = pk_bf16(v[2], v[3]); w[2] = pk_bf16(v[4], v[5]); w[3] = pk_bf16(v[6], v[7]);
    return __builtin_bit_cast(bf16x8, w);
}

// KS x (32 x 32) fp32 tiles -> bf16 LDS images [part][row][AB_LD], loaded by the whole workgroup (NT threads).
// Part p of the workgroup streams its own contiguous run of `nth` tiles (tile index p * nth + it): the in-workgroup split
// of the streamed dimension (see the forward kernel).  LPT = float4 per thread, operand and iteration.
template <int NT, int KS>
struct TileP {
    static constexpr int LPT = (KS * 256 + NT - 1) / NT;
    static_assert((KS * 256) % NT == 0, "the tiles of an iteration divide evenly over the threads");
    float4 v[LPT];
    // rows past nrows, and tiles past this part's run, take the out-of-range offset (zeros): one unconditional request each
    __device__ __forceinline__ void loadb(const BufSrc &src, unsigned ld_bytes, int it, int nth, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int u = tid + NT * i;
            const int r = u >> 3, c = (u & 7) * 4;
            const int row = ((r >> 5) * nth + it) * AT_KEYS + (r & 31);
            v[i] = src.ld4((it < nth && row < nrows) ? (unsigned)row * ld_bytes + 4u * (unsigned)c : BUF_OOB);
        }
    }
    __device__ __forceinline__ void store(unsigned short (*S)[2][AT_KEYS][AB_LD], int buf, int tid, float scale = 1.0f) const {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int u = tid + NT * i;
            const int r = u >> 3, c = (u & 7) * 4;
            *reinterpret_cast<uint2 *>(&S[r >> 5][buf][r & 31][c]) =
                make_uint2(pk_bf16(v[i].x * scale, v[i].y * scale), pk_bf16(v[i].z * scale, v[i].w * scale));
        }
    }
};

// row fragment: 8 consecutive columns (16*s + 8*hi ..) of tile row (lane & 31)
__device__ __forceinline__ bf16x8 frag_row(const unsigned short (*S)[AB_LD], int s, int lane) {
    return *reinterpret_cast<const bf16x8 *>(&S[lane & 31][16 * s + 8 * (lane >> 5)]);
}
// transposed fragment for k-step s2: lane (column c = lane & 31 of the tile, half hi) receives the 8 tile rows
// 16*s2 + 4*hi + {0,1,2,3, 8,9,10,11} of its column
__device__ __forceinline__ bf16x8 frag_col(const unsigned short (*S)[AB_LD], int s2, int lane) {
    const int g = lane >> 4, t = lane & 15;
    const int row = 16 * s2 + 4 * (g >> 1) + (t >> 2);
    const int col = 16 * (g & 1) + 4 * (t & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)&S[row][col]);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)&S[row + 8][col]);
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

#define MFMA_BF16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)

// 8 consecutive floats of one row through a buffer descriptor: two unconditional 16-byte requests (a lane outside the
// tensor takes the out-of-range offset).  The per-element `ok ? p[i] : 0` form compiled to 16 dependent round trips in
// the prologue of every workgroup -- about as long as the workgroup's whole key loop at S = 1050.
__device__ __forceinline__ void ld_row8(const BufSrc &src, bool ok, unsigned byte_off, float (&t)[8]) {
    const float4 x = src.ld4(ok ? byte_off : BUF_OOB), y = src.ld4(ok ? byte_off + 16u : BUF_OOB);
    t[0] = x.x; t[1] = x.y; t[2] = x.z; t[3] = x.w;
    t[4] = y.x; t[5] = y.y; t[6] = y.z; t[7] = y.w;
}

// ------------------------------------------------------------------------------------------------
// forward
//
// Workgroup = NW query waves x KS parts.  The encoder's 1050 x 1050 problems offer only ~2 waves per SIMD when a wave
// owns 32 queries and streams all keys (64 (b, h) problems x 33 query tiles / 1024 SIMDs), and the decoder's cross
// attention (100 queries x 1050 keys) a quarter of a wave: the kernels are bound by the latency of ONE wave's dependent
// chain per tile (MFMA -> row maximum -> exponentials -> row sum -> pack -> MFMA), which nothing else on the SIMD covers.
// With KS > 1 the key range is cut into KS contiguous runs; wave (qw, kp) streams run kp for the queries of qw with its own
// running (m, l, O), and the KS partial results are merged through LDS at the end (the log-sum-exp merge of two softmax
// partials).  Same arithmetic per key tile, twice / four times the waves in flight.
// ------------------------------------------------------------------------------------------------
template <int NW, int KS>
__global__ __launch_bounds__(64 * NW * KS, (KS == 2) ? 4 : 1) void attn_fwd_bf16_kernel(AttnArgs a) {
    constexpr int NT = 64 * NW * KS;
    __shared__ __attribute__((aligned(16))) unsigned short KVs[2 * KS][2][AT_KEYS][AB_LD];     // [K parts | V parts][buffer]
    unsigned short (*Ks)[2][AT_KEYS][AB_LD] = KVs, (*Vs)[2][AT_KEYS][AB_LD] = KVs + KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qw = wave % NW, kp = wave / NW;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int tq = blockIdx.x * (32 * NW) + qw * 32 + l31;
    const bool qok = tq < a.T;
    const float *Qb = a.Q + (long long)b * a.T * a.ldq + h * 32;
    const float *Kb = a.K + (long long)b * a.S * a.ldk + h * 32;
    const float *Vb = a.V + (long long)b * a.S * a.ldv + h * 32;
    const float qmul = a.qscale * AT_LOG2E;
    const uint32_t dkey = a.drop_scale != 0.0f ? drop_key(a.drop_seed, a.drop_step) : 0u;

    bf16x8 qb[2];                 // B operand of QK^T: Q[tq][16 s + 8 hi ..], in log2 units
    {
        BufSrc qsrc;
        qsrc.init(Qb, (long long)(a.T - 1) * a.ldq + 32);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float t[8];
            ld_row8(qsrc, qok, (unsigned)(((long long)tq * a.ldq + 16 * s + 8 * hi) * 4), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] *= qmul;
            qb[s] = pack8(t);
        }
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
    float m = -INFINITY, lsum = 0.0f;
    const unsigned long long rowbase = ((unsigned long long)bh * a.T + tq) * (unsigned long long)((a.S + 1) & ~1);
    const float lg2scale = a.drop_scale != 0.0f ? __log2f(a.drop_scale) : 0.0f;

    const int ntiles = (a.S + AT_KEYS - 1) / AT_KEYS;
    const int nth = (ntiles + KS - 1) / KS;            // tiles per part (the last part may own fewer)
    // K / V tiles: operand pipeline two tiles deep with LDS-only barriers (see gemm_bf16c_body in gemm_f32.hip): tile it
    // is consumed from LDS while tile it+1 waits in one register set and tile it+2 is in flight into the other.
    BufSrc ksrc, vsrc;
    ksrc.init(Kb, (long long)(a.S - 1) * a.ldk + 32);
    vsrc.init(Vb, (long long)(a.S - 1) * a.ldv + 32);
    const unsigned ldbk = (unsigned)(a.ldk * 4), ldbv = (unsigned)(a.ldv * 4);
    TileP<NT, KS> rk0, rv0, rk1, rv1;
    rk0.loadb(ksrc, ldbk, 0, nth, a.S, tid);
    rv0.loadb(vsrc, ldbv, 0, nth, a.S, tid);
    rk0.store(Ks, 0, tid);
    rv0.store(Vs, 0, tid);
    rk0.loadb(ksrc, ldbk, 1, nth, a.S, tid);
    rv0.loadb(vsrc, ldbv, 1, nth, a.S, tid);
    rk1.loadb(ksrc, ldbk, 2, nth, a.S, tid);
    rv1.loadb(vsrc, ldbv, 2, nth, a.S, tid);
    lds_barrier();
    auto tile = [&](const int it, const int cur, TileP<NT, KS> &rpk, TileP<NT, KS> &rpv) {
        rpk.store(Ks, cur ^ 1, tid);
        rpv.store(Vs, cur ^ 1, tid);
        rpk.loadb(ksrc, ldbk, it + 3, nth, a.S, tid);
        rpv.loadb(vsrc, ldbv, it + 3, nth, a.S, tid);
        const int kbase = (kp * nth + it) * AT_KEYS;
        if (kbase < a.S) {                          // (wave-uniform: the last part may run out of keys early)
        const unsigned short (*Kt)[AB_LD] = Ks[kp][cur];
        const unsigned short (*Vt)[AB_LD] = Vs[kp][cur];
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
        s = MFMA_BF16(frag_row(Kt, 0, lane), qb[0], s);
        s = MFMA_BF16(frag_row(Kt, 1, lane), qb[1], s);
        if (kbase + AT_KEYS > a.S) {                // ragged last tile only (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + krow(r, hi) >= a.S) s[r] = -INFINITY;
        }
        const float mx = halves_max(tree_max16(s));
        const float mn = fmaxf(m, mx);
        const float corr = fast_exp2(m - mn);
        // the dropout scale 1/(1-p) rides in the exponent (p_r / (1-p) = exp2(s_r - mn + log2 1/(1-p))): no multiply per kept
        // element; the row sums carry the same factor, which the epilogue takes out again
        const float mne = mn - lg2scale;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = fast_exp2(s[r] - mne);
        const float rs = halves_sum(tree_sum16(p));
        lsum = lsum * corr + rs;
        m = mn;
        if (a.drop_scale != 0.0f)
            drop_keep16(dkey, rowbase, kbase, hi, a.drop_thresh, [&](int r, bool keep) { p[r] = keep ? p[r] : 0.0f; });
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= corr;
        o = MFMA_BF16(frag_col(Vt, 0, lane), pack8(p), o);
        o = MFMA_BF16(frag_col(Vt, 1, lane), pack8(p + 8), o);
        }
        lds_barrier();
    };
    {
        int it = 0;
        for (; it + 2 <= nth; it += 2) {
            tile(it, 0, rk0, rv0);
            tile(it + 1, 1, rk1, rv1);
        }
        if (it < nth) tile(it, 0, rk0, rv0);
    }
    if constexpr (KS > 1) {
        // merge the KS partial softmaxes of a query: (m, l, O) of parts 1 .. KS-1 travel through LDS (component-major:
        // conflict-free), part 0 rescales everything to the common maximum.  A part without keys left (m = -inf, l = 0).
        __syncthreads();                                  // (every tile consumed, the trailing prefetches have landed in registers only)
        float *comb = reinterpret_cast<float *>(&KVs[0][0][0][0]);
        static_assert((KS - 1) * NW * 18 * 64 * 4 <= (int)sizeof(KVs), "merge buffer fits the tile buffers");
        if (kp > 0) {
            float *c = comb + ((kp - 1) * NW + qw) * 18 * 64 + lane;
            c[0] = m; c[64] = lsum;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[(2 + r) * 64] = o[r];
        }
        __syncthreads();
        if (kp > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k) {
            const float *c = comb + ((k - 1) * NW + qw) * 18 * 64 + lane;
            const float mb = c[0], lb = c[64];
            const float mn = fmaxf(m, mb);
            const float ca = fast_exp2(m - mn), cb = fast_exp2(mb - mn);
            lsum = lsum * ca + lb * cb;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = o[r] * ca + c[(2 + r) * 64] * cb;
            m = mn;
        }
    }
    if (qok) {
        // lsum carries the dropout scale (see mne above): the true row sum is lsum / scale
        const float dsc = a.drop_scale != 0.0f ? a.drop_scale : 1.0f;
        const float inv = dsc / lsum;
        float *Ob = a.O + ((long long)b * a.T + tq) * a.ldo + h * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) Ob[krow(r, hi)] = o[r] * inv;
        if (hi == 0) a.LSE[(long long)bh * a.T + tq] = m * AT_LN2 + logf(lsum / dsc);
    }
}

// ------------------------------------------------------------------------------------------------
// backward 1/2: dQ (per query tile, streams the keys; KS parts as in the forward, partial dQ summed through LDS)
// and delta = rowsum(dO * O)
// ------------------------------------------------------------------------------------------------
template <int NW, int KS>
__global__ __launch_bounds__(64 * NW * KS, (NW * KS > 4) ? 2 : 3) void attn_bwd_dq_bf16_kernel(AttnArgs a) {
    constexpr int NT = 64 * NW * KS;
    __shared__ __attribute__((aligned(16))) unsigned short KVs[2 * KS][2][AT_KEYS][AB_LD];
    unsigned short (*Ks)[2][AT_KEYS][AB_LD] = KVs, (*Vs)[2][AT_KEYS][AB_LD] = KVs + KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qw = wave % NW, kp = wave / NW;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int tq = blockIdx.x * (32 * NW) + qw * 32 + l31;
    const bool qok = tq < a.T;
    const long long qoff = ((long long)b * a.T + tq) * a.lddq + h * 32;
    const float *Kb = a.K + (long long)b * a.S * a.ldk + h * 32;
    const float *Vb = a.V + (long long)b * a.S * a.ldv + h * 32;
    const float qmul = a.qscale * AT_LOG2E;
    const uint32_t dkey = a.drop_scale != 0.0f ? drop_key(a.drop_seed, a.drop_step) : 0u;

    bf16x8 qb[2], dob[2];
    float dl = 0.0f;
    {
        BufSrc qsrc, dosrc, osrc;
        qsrc.init(a.Q + (long long)b * a.T * a.ldq + h * 32, (long long)(a.T - 1) * a.ldq + 32);
        dosrc.init(a.dO + (long long)b * a.T * a.lddo + h * 32, (long long)(a.T - 1) * a.lddo + 32);
        osrc.init(a.O + (long long)b * a.T * a.ldo + h * 32, (long long)(a.T - 1) * a.ldo + 32);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float tqv[8], tdo[8], tov[8];
            const unsigned cb = (unsigned)(16 * s + 8 * hi) * 4u;
            ld_row8(qsrc, qok, (unsigned)((long long)tq * a.ldq * 4) + cb, tqv);
            ld_row8(dosrc, qok, (unsigned)((long long)tq * a.lddo * 4) + cb, tdo);
            ld_row8(osrc, qok, (unsigned)((long long)tq * a.ldo * 4) + cb, tov);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                tqv[j] *= qmul;
                dl += tdo[j] * tov[j];
            }
            qb[s] = pack8(tqv);
            dob[s] = pack8(tdo);
        }
    }
    dl = halves_sum(dl);
    const float lse = qok ? a.LSE[(long long)bh * a.T + tq] * AT_LOG2E : INFINITY;
    if (qok && hi == 0 && kp == 0) a.delta[(long long)bh * a.T + tq] = dl;
    const unsigned long long rowbase = ((unsigned long long)bh * a.T + tq) * (unsigned long long)((a.S + 1) & ~1);

    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.0f;

    const int ntiles = (a.S + AT_KEYS - 1) / AT_KEYS;
    const int nth = (ntiles + KS - 1) / KS;
    BufSrc ksrc, vsrc;
    ksrc.init(Kb, (long long)(a.S - 1) * a.ldk + 32);
    vsrc.init(Vb, (long long)(a.S - 1) * a.ldv + 32);
    const unsigned ldbk = (unsigned)(a.ldk * 4), ldbv = (unsigned)(a.ldv * 4);
    TileP<NT, KS> rk0, rv0, rk1, rv1;
    rk0.loadb(ksrc, ldbk, 0, nth, a.S, tid);
    rv0.loadb(vsrc, ldbv, 0, nth, a.S, tid);
    rk0.store(Ks, 0, tid);
    rv0.store(Vs, 0, tid);
    rk0.loadb(ksrc, ldbk, 1, nth, a.S, tid);
    rv0.loadb(vsrc, ldbv, 1, nth, a.S, tid);
    rk1.loadb(ksrc, ldbk, 2, nth, a.S, tid);
    rv1.loadb(vsrc, ldbv, 2, nth, a.S, tid);
    lds_barrier();
    auto tile = [&](const int it, const int cur, TileP<NT, KS> &rpk, TileP<NT, KS> &rpv) {
        rpk.store(Ks, cur ^ 1, tid);
        rpv.store(Vs, cur ^ 1, tid);
        rpk.loadb(ksrc, ldbk, it + 3, nth, a.S, tid);
        rpv.loadb(vsrc, ldbv, it + 3, nth, a.S, tid);
        const int kbase = (kp * nth + it) * AT_KEYS;
        if (kbase < a.S) {
        const unsigned short (*Kt)[AB_LD] = Ks[kp][cur];
        const unsigned short (*Vt)[AB_LD] = Vs[kp][cur];
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            s = MFMA_BF16(frag_row(Kt, st, lane), qb[st], s);
            dp = MFMA_BF16(frag_row(Vt, st, lane), dob[st], dp);
        }
        float ds[16];
        // ds = p * (drop(dp) - delta), drop(dp) = keep ? dp / (1-p) : 0: one fused multiply-subtract per element either way
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(s[r] - lse);
        if (a.drop_scale != 0.0f) {
            const float ndl = -dl;
            drop_keep16(dkey, rowbase, kbase, hi, a.drop_thresh,
                        [&](int r, bool keep) { ds[r] *= keep ? __builtin_fmaf(dp[r], a.drop_scale, ndl) : ndl; });
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] *= dp[r] - dl;
        }
        if (kbase + AT_KEYS > a.S) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + krow(r, hi) >= a.S) ds[r] = 0.0f;
        }
        dq = MFMA_BF16(frag_col(Kt, 0, lane), pack8(ds), dq);
        dq = MFMA_BF16(frag_col(Kt, 1, lane), pack8(ds + 8), dq);
        }
        lds_barrier();
    };
    {
        int it = 0;
        for (; it + 2 <= nth; it += 2) {
            tile(it, 0, rk0, rv0);
            tile(it + 1, 1, rk1, rv1);
        }
        if (it < nth) tile(it, 0, rk0, rv0);
    }
    if constexpr (KS > 1) {
        __syncthreads();
        float *comb = reinterpret_cast<float *>(&KVs[0][0][0][0]);
        static_assert((KS - 1) * NW * 16 * 64 * 4 <= (int)sizeof(KVs), "merge buffer fits the tile buffers");
        if (kp > 0) {
            float *c = comb + ((kp - 1) * NW + qw) * 16 * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r * 64] = dq[r];
        }
        __syncthreads();
        if (kp > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k) {
            const float *c = comb + ((k - 1) * NW + qw) * 16 * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[r] += c[r * 64];
        }
    }
    if (qok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a.dQ[qoff + krow(r, hi)] = dq[r] * a.qscale;     // gradient w.r.t. the unscaled q
    }
}

// ------------------------------------------------------------------------------------------------
// backward 2/2: dK, dV (per key tile, streams the queries): lane l holds key (l&31) and 16 queries krow(r, hi).
// The Q tile is staged as bf16(Q * log2 e) -- the forward's rounded operand -- and K unscaled.
// KS parts as in the forward, here over the QUERY tiles: wave (kw, qp) accumulates dK / dV of the keys of kw over the
// query run qp; the partial sums are added through LDS at the end (fixed order: deterministic).
// ------------------------------------------------------------------------------------------------
template <int NW, int KS>
__global__ __launch_bounds__(64 * NW * KS, (NW * KS > 4) ? 2 : 3) void attn_bwd_dkv_bf16_kernel(AttnArgs a) {
    constexpr int NT = 64 * NW * KS;
    __shared__ __attribute__((aligned(16))) unsigned short QDs[2 * KS][2][AT_KEYS][AB_LD];     // [Q parts | dO parts][buffer]
    __shared__ __attribute__((aligned(16))) float Ls[KS][2][AT_KEYS], Dl[KS][2][AT_KEYS];
    unsigned short (*Qs)[2][AT_KEYS][AB_LD] = QDs, (*Ds)[2][AT_KEYS][AB_LD] = QDs + KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kw = wave % NW, qp = wave / NW;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int sk = blockIdx.x * (32 * NW) + kw * 32 + l31;
    const bool kok = sk < a.S;
    const long long dkoff = ((long long)b * a.S + sk) * a.lddk + h * 32, dvoff = ((long long)b * a.S + sk) * a.lddv + h * 32;
    const float *Qb = a.Q + (long long)b * a.T * a.ldq + h * 32;
    const float *Db = a.dO + (long long)b * a.T * a.lddo + h * 32;
    const float qmul = a.qscale * AT_LOG2E;
    const uint32_t dkey = a.drop_scale != 0.0f ? drop_key(a.drop_seed, a.drop_step) : 0u;
    const float *lse = a.LSE + (long long)bh * a.T;
    const float *dlt = a.delta + (long long)bh * a.T;
    const unsigned long long Sp = (unsigned long long)((a.S + 1) & ~1);

    bf16x8 kb[2], vb[2];
    {
        BufSrc ksrc, vsrc;
        ksrc.init(a.K + (long long)b * a.S * a.ldk + h * 32, (long long)(a.S - 1) * a.ldk + 32);
        vsrc.init(a.V + (long long)b * a.S * a.ldv + h * 32, (long long)(a.S - 1) * a.ldv + 32);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float tk[8], tv[8];
            const unsigned cb = (unsigned)(16 * s + 8 * hi) * 4u;
            ld_row8(ksrc, kok, (unsigned)((long long)sk * a.ldk * 4) + cb, tk);
            ld_row8(vsrc, kok, (unsigned)((long long)sk * a.ldv * 4) + cb, tv);
            kb[s] = pack8(tk);
            vb[s] = pack8(tv);
        }
    }
    f32x16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[r] = 0.0f; dv[r] = 0.0f; }

    const int ntiles = (a.T + AT_KEYS - 1) / AT_KEYS;
    const int nth = (ntiles + KS - 1) / KS;           // query tiles per part
    constexpr int LPP = (KS * AT_KEYS + NT - 1) / NT; // (lse, delta) pairs a thread stages per iteration (0 or 1 here)
    // (one-tile-deep staging here: the two-deep form of the forward / dQ kernels needs a second register set, which takes
    //  this kernel from 2 waves per SIMD to 1 -- measured +0.7 ms per step)
    BufSrc qsrc, dosrc, lsrc, dlsrc;
    qsrc.init(Qb, (long long)(a.T - 1) * a.ldq + 32);
    dosrc.init(Db, (long long)(a.T - 1) * a.lddo + 32);
    lsrc.init(lse, a.T);
    dlsrc.init(dlt, a.T);
    const unsigned ldbq = (unsigned)(a.ldq * 4), ldbd = (unsigned)(a.lddo * 4);
    // With dropout the scale 1/(1-p) rides in the exponent, as in the forward: the staged row constants are
    //   L' = lse * log2 e - log2 scale   (exp2(s - L') = P * scale)      and      delta' = delta / scale,
    // so that  dS = P (drop(dP) - delta) = (P scale) ((keep ? dP : 0) - delta')  and  drop(P) = keep ? P scale : 0.
    const float lg2scale = a.drop_scale != 0.0f ? __log2f(a.drop_scale) : 0.0f;
    const float inv_scale = a.drop_scale != 0.0f ? 1.0f / a.drop_scale : 1.0f;
    TileP<NT, KS> rq, rd;
    float rl[LPP], rdl[LPP];
    // raw (lse, delta) of iteration `it` for the rows this thread stages: row t of the iteration = part t >> 5, query
    // (part * nth + it) * 32 + (t & 31); scaling / padding happens when the set is stored (arithmetic on a just-requested
    // value would put a vmcnt wait in front of the iteration's MFMAs)
    auto load_rows = [&](int it) {
#pragma unroll
        for (int j = 0; j < LPP; ++j) {
            const int t = tid + NT * j;
            const int q = ((t >> 5) * nth + it) * AT_KEYS + (t & 31);
            const bool ok = t < KS * AT_KEYS && it < nth && q < a.T;
            rl[j] = lsrc.ld1(ok ? 4u * (unsigned)q : BUF_OOB);
            rdl[j] = dlsrc.ld1(ok ? 4u * (unsigned)q : BUF_OOB);
        }
    };
    auto store_rows = [&](int it, int buf) {
#pragma unroll
        for (int j = 0; j < LPP; ++j) {
            const int t = tid + NT * j;
            if (t < KS * AT_KEYS) {
                const int q = ((t >> 5) * nth + it) * AT_KEYS + (t & 31);
                const bool ok = it < nth && q < a.T;
                Ls[t >> 5][buf][t & 31] = ok ? rl[j] * AT_LOG2E - lg2scale : INFINITY;     // +inf => p = exp2(-inf) = 0 for padded queries
                Dl[t >> 5][buf][t & 31] = ok ? rdl[j] * inv_scale : 0.0f;
            }
        }
    };
    rq.loadb(qsrc, ldbq, 0, nth, a.T, tid);
    rd.loadb(dosrc, ldbd, 0, nth, a.T, tid);
    load_rows(0);
    rq.store(Qs, 0, tid, qmul);      // bf16(Q * log2 e): the SAME rounded operands as the forward / dQ kernels, so that
    rd.store(Ds, 0, tid);            // exp2(s - lse) is consistent with the stored LSE; dK is rescaled by ln 2 at the end
    store_rows(0, 0);
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < nth; ++it) {
        const bool more = (it + 1) < nth;
        if (more) {
            rq.loadb(qsrc, ldbq, it + 1, nth, a.T, tid);
            rd.loadb(dosrc, ldbd, it + 1, nth, a.T, tid);
            load_rows(it + 1);
        }
        const int qbase = (qp * nth + it) * AT_KEYS;
        if (qbase < a.T) {                          // (wave-uniform: the last part may run out of queries early)
        const unsigned short (*Qt)[AB_LD] = Qs[qp][cur];
        const unsigned short (*Dt)[AB_LD] = Ds[qp][cur];
        const float *Lt = Ls[qp][cur], *Dlt = Dl[qp][cur];
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            s = MFMA_BF16(frag_row(Qt, st, lane), kb[st], s);
            dp = MFMA_BF16(frag_row(Dt, st, lane), vb[st], dp);
        }
        float p[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = fast_exp2(s[r] - Lt[krow(r, hi)]);          // P (* scale with dropout)
        if (a.drop_scale != 0.0f) {
            drop_keep16_keycol(dkey, (unsigned long long)bh * a.T + qbase, Sp, sk, lane, hi, a.drop_thresh, [&](int r, bool keep) {
                ds[r] = p[r] * ((keep ? dp[r] : 0.0f) - Dlt[krow(r, hi)]);
                p[r] = keep ? p[r] : 0.0f;                                      // dV uses the dropped probabilities
            });
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = p[r] * (dp[r] - Dlt[krow(r, hi)]);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            dv = MFMA_BF16(frag_col(Dt, s2, lane), pack8(p + 8 * s2), dv);
            dk = MFMA_BF16(frag_col(Qt, s2, lane), pack8(ds + 8 * s2), dk);
        }
        }
        if (more) {
            rq.store(Qs, cur ^ 1, tid, qmul);
            rd.store(Ds, cur ^ 1, tid);
            store_rows(it + 1, cur ^ 1);
        }
        __syncthreads();
        cur ^= 1;
    }
    if constexpr (KS > 1) {
        float *comb = reinterpret_cast<float *>(&QDs[0][0][0][0]);
        static_assert((KS - 1) * NW * 32 * 64 * 4 <= (int)sizeof(QDs), "merge buffer fits the tile buffers");
        if (qp > 0) {
            float *c = comb + ((qp - 1) * NW + kw) * 32 * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) { c[r * 64] = dk[r]; c[(16 + r) * 64] = dv[r]; }
        }
        __syncthreads();
        if (qp > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k) {
            const float *c = comb + ((k - 1) * NW + kw) * 32 * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[r] += c[r * 64]; dv[r] += c[(16 + r) * 64]; }
        }
    }
    if (kok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a.dK[dkoff + krow(r, hi)] = dk[r] * AT_LN2;      // (the staged Q carries qscale * log2 e)
            a.dV[dvoff + krow(r, hi)] = dv[r];
        }
    }
}

}  // namespace detr

namespace detr {

// Parts of the streamed dimension per workgroup (KS): enough waves to cover each other's dependent chains -- the aim is
// >= 4 waves per SIMD (4096 on the chip) -- but at least 8 streamed tiles per part.  DETR_HIP_ATTN_SPLIT=1 / 2 / 4 forces.
static int attn_parts(int rows, int streamed, int bh, int max_parts) {
    // Measured (scripts/micro_attn.py, dropout 0.1, us fwd / bwd; 1 / 2 / 4 parts): B8 1050x1050 66 / 140, 47.5 / 124, 52 / 151;
    // B16 1050x1050 98 / 222, 78 / 206, 90 / 272; B8 100x1050 (cross attention) 33 / 54, 17 / 43, 16 / 43; B8 1344x1344
    // 79 / 173, 65 / 170, 72 / 206.  (The two-part backward kernels squeezed to 128 VGPRs for 4 waves per SIMD spill 33 dwords
    // and run 1.9x slower: they stay at 3.)
    const int force = tune(T_ATTN_SPLIT);
    if (force == 1 || force == 2) return force;
    if (force == 4) return max_parts >= 4 ? 4 : 2;
    const long long waves = (long long)cdiv(rows, 64) * 2 * bh;       // 2-wave row groups
    const int tiles = cdiv(streamed, AT_KEYS);
    if (waves >= 8192 || tiles < 16) return 1;
    if (waves * 2 >= 2048 || tiles < 32 || max_parts < 4) return 2;
    return 4;
}

int attn_fwd_bf16_launch(const AttnArgs &a, hipStream_t s) {
    const int bh = a.B * a.H;
    if (attn_waves(a.T, bh) == 2) {
        const dim3 grid((unsigned)cdiv(a.T, 64), (unsigned)bh);
        const int ks = attn_parts(a.T, a.S, bh, 4);
        if (ks == 4) hipLaunchKernelGGL((attn_fwd_bf16_kernel<2, 4>), grid, dim3(512), 0, s, a);
        else if (ks == 2) hipLaunchKernelGGL((attn_fwd_bf16_kernel<2, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_fwd_bf16_kernel<2, 1>), grid, dim3(128), 0, s, a);
    } else hipLaunchKernelGGL((attn_fwd_bf16_kernel<4, 1>), dim3((unsigned)cdiv(a.T, 128), (unsigned)bh), dim3(256), 0, s, a);
    DETR_LAUNCH_CHECK("attention fwd (bf16 MFMA)");
    return 0;
}

int attn_bwd_bf16_launch(const AttnArgs &a, hipStream_t s) {
    const int bh = a.B * a.H;
    if (attn_waves(a.T, bh) == 2) {
        const dim3 grid((unsigned)cdiv(a.T, 64), (unsigned)bh);
        const int ks = attn_parts(a.T, a.S, bh, 4);
        if (ks == 4) hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<2, 4>), grid, dim3(512), 0, s, a);
        else if (ks == 2) hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<2, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<2, 1>), grid, dim3(128), 0, s, a);
    } else hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<4, 1>), dim3((unsigned)cdiv(a.T, 128), (unsigned)bh), dim3(256), 0, s, a);
    DETR_LAUNCH_CHECK("attention bwd dq (bf16 MFMA)");
    if (attn_waves(a.S, bh) == 2) {
        const dim3 grid((unsigned)cdiv(a.S, 64), (unsigned)bh);
        if (attn_parts(a.S, a.T, bh, 2) >= 2) hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<2, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<2, 1>), grid, dim3(128), 0, s, a);
    } else hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<4, 1>), dim3((unsigned)cdiv(a.S, 128), (unsigned)bh), dim3(256), 0, s, a);
    DETR_LAUNCH_CHECK("attention bwd dkv (bf16 MFMA)");
    return 0;
}

}  // namespace detr
