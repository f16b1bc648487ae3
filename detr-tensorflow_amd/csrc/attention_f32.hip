// attention_f32.hip -- fused multi-head attention core (head_dim 32, fp32) for gfx950:
//     O = softmax(Q K^T) V        (Q is already scaled by 1/sqrt(32), transformer.py:307,317,340-343)
// and its backward, flash-style: the [T, S] score / probability tensor (283 MB per encoder layer
// at B=8, 800x1333) never touches HBM.  Layout: batch-first token matrices [B, T, heads*32]
// (row stride ld floats); (batch, head) select a column block of 32.
//
// MFMA formulation (v_mfma_f32_32x32x2_f32, exact f32).  Everything is computed TRANSPOSED so that
// the softmax row (one query) lives in ONE lane pair (l, l^32):
//     S^T[key][query] = K_tile (A: rows = keys, LDS) x Q^T (B: per-lane registers)
//   C/D map: lane l holds query (l&31) and the 16 keys  kr = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15
//   -> row max / row sum = 15 VALU ops + one cross-half shuffle (no LDS, no serial lanes).
//     O^T[d][query]  += V^T (A: V[key][d] read row-wise from LDS) x P^T (B: the registers above;
//   MFMA k-step r contracts the key pair {kr(hi=0), kr(hi=1)} which is exactly what the two lane
//   halves hold in register r -- no data movement between QK^T and PV).
// Softmax arithmetic: Q (or, in the dK/dV kernel, K) is pre-multiplied by log2(e) in registers so that every
// exponential is one v_exp_f32 (exp2); LSE is stored in natural-log units.  The key-bound masks run only in the
// last key tile (wave-uniform branch), and the dropout hash is evaluated once per PAIR of adjacent keys
// (common.h drop_hash: 16 bits per element; element index = row * Sp + key, Sp = S rounded up to even).
// K/V tiles: 32 keys x 32 dims row-major with row stride 33 dwords: conflict-free for both access
// patterns (lanes over keys at fixed d, and lanes over d at fixed key).  Double-buffered
// global->register->LDS pipeline, one barrier per key tile; 4 waves x 32 queries per workgroup.
#include "attention_common.h"
#include "gemm_core.h"        // BufSrc: buffer-descriptor loads

namespace detr {

// Elements hi, 2 + hi, 4 + hi, ... of one 32-float head row (the 32x32x2 MFMA operand map of a lane): the row is fetched with
// eight unconditional 16-byte buffer-descriptor requests and the lane picks its half -- the per-element `ok ? p[i] : 0`
// form compiled to 16 dependent round trips per tensor in the prologue of every workgroup.
__device__ __forceinline__ void ld_row_stride2(const float *base, long long extent, bool ok, long long row_off, int hi, float (&out)[16]) {
    BufSrc src;
    src.init(base, extent);
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src.ld4(ok ? (unsigned)(row_off * 4) + 16u * i : BUF_OOB);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        out[2 * i] = hi ? v[i].y : v[i].x;
        out[2 * i + 1] = hi ? v[i].w : v[i].z;
    }
}


constexpr int AT_LD = 33;

// cooperative load of one 32 x 32 tile (rows row0.., zero filled past nrows) into registers / LDS by a workgroup of
// NW waves: 256 float4 per tile, 4 / NW per thread
template <int NW>
struct Tile {
    static constexpr int LPT = 4 / NW;
    float4 v[LPT];
    __device__ __forceinline__ void load(const float *base, long long ld, int row0, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int u = tid + 64 * NW * i;
            const int r = u >> 3, c = (u & 7) * 4;
            const int row = row0 + r;
            v[i] = (row < nrows) ? *reinterpret_cast<const float4 *>(base + (long long)row * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __device__ __forceinline__ void store(float (*S)[AT_LD], int tid) const {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int u = tid + 64 * NW * i;
            const int r = u >> 3, c = (u & 7) * 4;
            S[r][c + 0] = v[i].x; S[r][c + 1] = v[i].y; S[r][c + 2] = v[i].z; S[r][c + 3] = v[i].w;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(AttnArgs a) {
    __shared__ float Ks[2][AT_KEYS][AT_LD];
    __shared__ float Vs[2][AT_KEYS][AT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int tq = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const bool qok = tq < a.T;
    const float *Qb = a.Q + (long long)b * a.T * a.ldq + h * 32;
    const float *Kb = a.K + (long long)b * a.S * a.ldk + h * 32;
    const float *Vb = a.V + (long long)b * a.S * a.ldv + h * 32;
    const float qmul = a.qscale * AT_LOG2E;
    const uint32_t dkey = a.drop_scale != 0.0f ? drop_key(a.drop_seed, a.drop_step) : 0u;

    float q[16];
    ld_row_stride2(Qb, (long long)(a.T - 1) * a.ldq + 32, qok, (long long)tq * a.ldq, hi, q);
#pragma unroll
    for (int s = 0; s < 16; ++s) q[s] *= qmul;                                                          // scores in log2 units

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
    float m = -INFINITY, lsum = 0.0f;
    const unsigned long long rowbase = ((unsigned long long)bh * a.T + tq) * (unsigned long long)((a.S + 1) & ~1);

    const int ntiles = (a.S + AT_KEYS - 1) / AT_KEYS;
    Tile<NW> rk, rv;
    rk.load(Kb, a.ldk, 0, a.S, tid);
    rv.load(Vb, a.ldv, 0, a.S, tid);
    rk.store(Ks[0], tid);
    rv.store(Vs[0], tid);
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < ntiles; ++it) {
        const bool more = (it + 1) < ntiles;
        if (more) {
            rk.load(Kb, a.ldk, (it + 1) * AT_KEYS, a.S, tid);
            rv.load(Vb, a.ldv, (it + 1) * AT_KEYS, a.S, tid);
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int st = 0; st < 16; ++st)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[cur][l31][2 * st + hi], q[st], s, 0, 0, 0);
        const int kbase = it * AT_KEYS;
        if (kbase + AT_KEYS > a.S) {                // ragged last tile only (wave-uniform)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + krow(r, hi) >= a.S) s[r] = -INFINITY;
        }
        const float mx = halves_max(tree_max16(s));
        const float mn = fmaxf(m, mx);              // finite: every tile holds at least one valid key
        const float corr = fast_exp2(m - mn);       // exp2(-inf) = 0 on the first tile
        float rs = 0.0f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = fast_exp2(s[r] - mn);
            rs += p[r];
        }
        rs += __shfl_xor(rs, 32, 64);
        lsum = lsum * corr + rs;
        m = mn;
        if (a.drop_scale != 0.0f) {          // dropout on the attention probabilities (after normalisation == on p)
            drop_keep16(dkey, rowbase, kbase, hi, a.drop_thresh, [&](int r, bool keep) { p[r] = keep ? p[r] * a.drop_scale : 0.0f; });
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= corr;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[cur][krow(r, hi)][l31], p[r], o, 0, 0, 0);
        if (more) {
            rk.store(Ks[cur ^ 1], tid);
            rv.store(Vs[cur ^ 1], tid);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (qok) {
        const float inv = 1.0f / lsum;
        float *Ob = a.O + ((long long)b * a.T + tq) * a.ldo + h * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) Ob[krow(r, hi)] = o[r] * inv;
        if (hi == 0) a.LSE[(long long)bh * a.T + tq] = m * AT_LN2 + logf(lsum);   // natural-log units
    }
}

// ------------------------------------------------------------------------------------------------
// backward 1/2: dQ (per query tile, streams the keys) and delta = rowsum(dO * O)
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(AttnArgs a) {
    __shared__ float Ks[2][AT_KEYS][AT_LD];
    __shared__ float Vs[2][AT_KEYS][AT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int tq = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const bool qok = tq < a.T;
    const long long qoff = ((long long)b * a.T + tq) * a.lddq + h * 32;
    const float *Kb = a.K + (long long)b * a.S * a.ldk + h * 32;
    const float *Vb = a.V + (long long)b * a.S * a.ldv + h * 32;
    const float qmul = a.qscale * AT_LOG2E;
    const uint32_t dkey = a.drop_scale != 0.0f ? drop_key(a.drop_seed, a.drop_step) : 0u;

    float q[16], dout[16];
    float dl = 0.0f;
    {
        float ov[16];
        ld_row_stride2(a.Q + (long long)b * a.T * a.ldq + h * 32, (long long)(a.T - 1) * a.ldq + 32, qok, (long long)tq * a.ldq, hi, q);
        ld_row_stride2(a.dO + (long long)b * a.T * a.lddo + h * 32, (long long)(a.T - 1) * a.lddo + 32, qok, (long long)tq * a.lddo, hi, dout);
        ld_row_stride2(a.O + (long long)b * a.T * a.ldo + h * 32, (long long)(a.T - 1) * a.ldo + 32, qok, (long long)tq * a.ldo, hi, ov);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            q[s] *= qmul;                                           // scores in log2 units (q feeds only s)
            dl += dout[s] * ov[s];
        }
    }
    dl += __shfl_xor(dl, 32, 64);
    const float lse = qok ? a.LSE[(long long)bh * a.T + tq] * AT_LOG2E : INFINITY;
    if (qok && hi == 0) a.delta[(long long)bh * a.T + tq] = dl;
    const unsigned long long rowbase = ((unsigned long long)bh * a.T + tq) * (unsigned long long)((a.S + 1) & ~1);

    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.0f;

    const int ntiles = (a.S + AT_KEYS - 1) / AT_KEYS;
    Tile<NW> rk, rv;
    rk.load(Kb, a.ldk, 0, a.S, tid);
    rv.load(Vb, a.ldv, 0, a.S, tid);
    rk.store(Ks[0], tid);
    rv.store(Vs[0], tid);
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < ntiles; ++it) {
        const bool more = (it + 1) < ntiles;
        if (more) {
            rk.load(Kb, a.ldk, (it + 1) * AT_KEYS, a.S, tid);
            rv.load(Vb, a.ldv, (it + 1) * AT_KEYS, a.S, tid);
        }
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[cur][l31][2 * st + hi], q[st], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[cur][l31][2 * st + hi], dout[st], dp, 0, 0, 0);
        }
        const int kbase = it * AT_KEYS;
        float ds[16];
        if (a.drop_scale != 0.0f) {
            drop_keep16(dkey, rowbase, kbase, hi, a.drop_thresh, [&](int r, bool keep) { dp[r] = keep ? dp[r] * a.drop_scale : 0.0f; });
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(s[r] - lse) * (dp[r] - dl);
        if (kbase + AT_KEYS > a.S) {                // ragged last tile only: keys past S contribute nothing
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + krow(r, hi) >= a.S) ds[r] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dq = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[cur][krow(r, hi)][l31], ds[r], dq, 0, 0, 0);
        if (more) {
            rk.store(Ks[cur ^ 1], tid);
            rv.store(Vs[cur ^ 1], tid);
        }
        __syncthreads();
        cur ^= 1;
    }
    if (qok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a.dQ[qoff + krow(r, hi)] = dq[r] * a.qscale;     // gradient w.r.t. the unscaled q
    }
}

// ------------------------------------------------------------------------------------------------
// backward 2/2: dK, dV (per key tile, streams the queries; needs LSE and delta)
// here the natural orientation is S[query][key]: lane l holds key (l&31) and 16 queries.
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkv_kernel(AttnArgs a) {
    __shared__ float Qs[2][AT_KEYS][AT_LD];
    __shared__ float Ds[2][AT_KEYS][AT_LD];
    __shared__ float Ls[2][AT_KEYS], Dl[2][AT_KEYS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int bh = blockIdx.y, b = bh / a.H, h = bh % a.H;
    const int sk = blockIdx.x * (32 * NW) + wave * 32 + l31;
    const bool kok = sk < a.S;
    const long long dkoff = ((long long)b * a.S + sk) * a.lddk + h * 32, dvoff = ((long long)b * a.S + sk) * a.lddv + h * 32;
    const float *Qb = a.Q + (long long)b * a.T * a.ldq + h * 32;
    const float *Db = a.dO + (long long)b * a.T * a.lddo + h * 32;
    const float qmul = a.qscale * AT_LOG2E;
    const uint32_t dkey = a.drop_scale != 0.0f ? drop_key(a.drop_seed, a.drop_step) : 0u;
    const float *lse = a.LSE + (long long)bh * a.T;
    const float *dlt = a.delta + (long long)bh * a.T;
    const unsigned long long Sp = (unsigned long long)((a.S + 1) & ~1);

    float kk[16], vv[16];
    {
        ld_row_stride2(a.K + (long long)b * a.S * a.ldk + h * 32, (long long)(a.S - 1) * a.ldk + 32, kok, (long long)sk * a.ldk, hi, kk);
        ld_row_stride2(a.V + (long long)b * a.S * a.ldv + h * 32, (long long)(a.S - 1) * a.ldv + 32, kok, (long long)sk * a.ldv, hi, vv);
#pragma unroll
        for (int s = 0; s < 16; ++s) kk[s] *= qmul;                // scores in log2 units (kk feeds only s)
    }
    f32x16 dk, dv;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[r] = 0.0f; dv[r] = 0.0f; }

    const int ntiles = (a.T + AT_KEYS - 1) / AT_KEYS;
    Tile<NW> rq, rd;
    rq.load(Qb, a.ldq, 0, a.T, tid);
    rd.load(Db, a.lddo, 0, a.T, tid);
    float rl = 0.f, rdl = 0.f;
    if (tid < AT_KEYS) {
        rl = (tid < a.T) ? lse[tid] * AT_LOG2E : INFINITY;      // +inf => p = exp2(-inf) = 0 for padded queries
        rdl = (tid < a.T) ? dlt[tid] : 0.0f;
    }
    rq.store(Qs[0], tid);
    rd.store(Ds[0], tid);
    if (tid < AT_KEYS) { Ls[0][tid] = rl; Dl[0][tid] = rdl; }
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < ntiles; ++it) {
        const bool more = (it + 1) < ntiles;
        if (more) {
            const int t0 = (it + 1) * AT_KEYS;
            rq.load(Qb, a.ldq, t0, a.T, tid);
            rd.load(Db, a.lddo, t0, a.T, tid);
            if (tid < AT_KEYS) {
                rl = (t0 + tid < a.T) ? lse[t0 + tid] * AT_LOG2E : INFINITY;
                rdl = (t0 + tid < a.T) ? dlt[t0 + tid] : 0.0f;
            }
        }
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[cur][l31][2 * st + hi], kk[st], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[cur][l31][2 * st + hi], vv[st], dp, 0, 0, 0);
        }
        float p[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = fast_exp2(s[r] - Ls[cur][krow(r, hi)]);
        if (a.drop_scale != 0.0f) {
            drop_keep16_keycol(dkey, (unsigned long long)bh * a.T + it * AT_KEYS, Sp, sk, lane, hi, a.drop_thresh, [&](int r, bool keep) {
                const float dpr = keep ? dp[r] * a.drop_scale : 0.0f;
                ds[r] = p[r] * (dpr - Dl[cur][krow(r, hi)]);
                p[r] = keep ? p[r] * a.drop_scale : 0.0f;       // dV uses the dropped probabilities
            });
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = p[r] * (dp[r] - Dl[cur][krow(r, hi)]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = krow(r, hi);
            dv = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[cur][qr][l31], p[r], dv, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[cur][qr][l31], ds[r], dk, 0, 0, 0);
        }
        if (more) {
            rq.store(Qs[cur ^ 1], tid);
            rd.store(Ds[cur ^ 1], tid);
            if (tid < AT_KEYS) { Ls[cur ^ 1][tid] = rl; Dl[cur ^ 1][tid] = rdl; }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (kok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a.dK[dkoff + krow(r, hi)] = dk[r] * a.qscale;     // the staged Q is unscaled (K carries qscale * log2 e)
            a.dV[dvoff + krow(r, hi)] = dv[r];
        }
    }
}

}  // namespace detr

using namespace detr;

/* one entry point per direction; detr_attn_desc.compute selects the exact-fp32 kernels above or the bf16-MFMA kernels of
 * attention_bf16.hip */
extern "C" int detr_hip_attention_fwd(const detr_attn_desc *d, void *stream) {
    DETR_REQUIRE(d, "attention: null descriptor");
    if (d->io_dtype == 1) return attn2_fwd_from_desc(d, (hipStream_t)stream);      // all-bf16 operands: attention_dma.hip
    DETR_REQUIRE(d->io_dtype == 0, "attention: io_dtype must be 0 (fp32 tensors) or 1 (bf16 tensors)");
    AttnArgs a;
    if (attn_from_desc(d, 0, a)) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (d->compute == 1) return attn_fwd_bf16_launch(a, s);
    if (attn_waves(a.T, a.B * a.H) == 2) hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3((unsigned)cdiv(a.T, 64), (unsigned)(a.B * a.H)), dim3(128), 0, s, a);
    else hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3((unsigned)cdiv(a.T, 128), (unsigned)(a.B * a.H)), dim3(256), 0, s, a);
    DETR_LAUNCH_CHECK("attention fwd");
    return 0;
}

extern "C" int detr_hip_attention_bwd(const detr_attn_desc *d, void *stream) {
    DETR_REQUIRE(d, "attention: null descriptor");
    if (d->io_dtype == 1) return attn2_bwd_from_desc(d, (hipStream_t)stream);
    DETR_REQUIRE(d->io_dtype == 0, "attention: io_dtype must be 0 (fp32 tensors) or 1 (bf16 tensors)");
    AttnArgs a;
    if (attn_from_desc(d, 1, a)) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (d->compute == 1) return attn_bwd_bf16_launch(a, s);
    if (attn_waves(a.T, a.B * a.H) == 2) hipLaunchKernelGGL(attn_bwd_dq_kernel<2>, dim3((unsigned)cdiv(a.T, 64), (unsigned)(a.B * a.H)), dim3(128), 0, s, a);
    else hipLaunchKernelGGL(attn_bwd_dq_kernel<4>, dim3((unsigned)cdiv(a.T, 128), (unsigned)(a.B * a.H)), dim3(256), 0, s, a);
    DETR_LAUNCH_CHECK("attention bwd dq");
    if (attn_waves(a.S, a.B * a.H) == 2) hipLaunchKernelGGL(attn_bwd_dkv_kernel<2>, dim3((unsigned)cdiv(a.S, 64), (unsigned)(a.B * a.H)), dim3(128), 0, s, a);
    else hipLaunchKernelGGL(attn_bwd_dkv_kernel<4>, dim3((unsigned)cdiv(a.S, 128), (unsigned)(a.B * a.H)), dim3(256), 0, s, a);
    DETR_LAUNCH_CHECK("attention bwd dkv");
    return 0;
}
