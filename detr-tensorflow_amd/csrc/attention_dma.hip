// attention_dma.hip -- round 6: the fused attention core for ALL-bf16 operands (detr_attn_desc.io_dtype = 1), rebuilt around the
// operator's real bound at head_dim 32 (detr_tf/networks/transformer.py:308-345).
//
// What bounds it.  A 32 x 32 score tile is 4 MFMAs (2 for K Q^T, 2 for P V: 128 cycles of the matrix pipe) and one softmax
// (16 scores per lane).  The kernels of attention_bf16.hip spent ~340 VALU issue slots on that tile -- 160 of them the dropout mask
// (8 counter hashes per lane with two quarter-rate 32-bit multiplies each), re-evaluated by all three kernels --, staged fp32 K / V
// through VGPRs with a bf16 packing pass in every workgroup that streamed them, and ran one workgroup barrier per 32 keys.  Here:
//   * operands are bf16 in memory (the projection GEMMs store them; Q carries scale * log2(e)), outputs bf16: no conversion pass,
//     half the bytes of every K / V re-stream;
//   * the dropout keep flags are BITS, produced once per step by attn2_dropmask_kernel (one launch per attention site, off the
//     critical path) in the two layouts the kernels want -- one word per (query, 32-key tile) with bit = key for the kernels whose
//     lanes own queries, one word per (key, 32-query tile) with bit = query for the dK / dV kernel -- so that an element's flag is
//     v_bfe_i32 + v_and_b32 on a word that arrived with the tile.  Same function as common.h::drop_keep, bit for bit;
//   * a WAVE is the unit of work: 64 queries (two independent 32-query chains that share every K / V fragment) x a contiguous run
//     of key tiles.  Its K / V / flag tiles arrive by LDS-DMA (buffer_load ... lds) in a wave-PRIVATE ring of three stages,
//     published by the wave's own counted s_waitcnt vmcnt(N): no workgroup barrier in the key loop at all.  The workgroup exists
//     only to merge the partial softmaxes of its `parts` waves through LDS at the end (fixed order: deterministic);
//   * the LDS image of a tile is chosen by the source address each lane requests (the DMA writes lane-linearly): 64-byte rows,
//     the 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3) -- conflict-free for the row fragments (ds_read_b128) and for the
//     transposed fragments (ds_read_b64_tr_b16), tests/test_dma_images_cpu.py enumerates both against the bank model;
//   * softmax: the running reference M only moves when a tile's maximum exceeds it by more than 2^6 (T13 of the guide; the first
//     tile sets it exactly), and -M (+ log2 of the dropout scale) is the C operand of the first QK^T MFMA, so the shifted score
//     comes out of the matrix pipe and an element costs exp2 + add-to-sum + 2 flag ops + half a cvt_pk;
//   * the backward keeps its two deterministic kernels (dQ per query block, dK / dV per key block) in the same form; the dQ kernel
//     leaves delta / scale and lse * log2(e) - log2(scale) in `stats` for the dK / dV kernel, which gets them by DMA with its tiles.
// Fragment maps (K Q^T transposed so that a score row lives in one lane pair, P^T already in the B layout of the PV product) are
// those of attention_bf16.hip.
#include "attention_common.h"
#include "gemm_ring.h"        // ring_dma_piece, ring_rsrc, ring_lds_addr, ring_wait_vmcnt
#include <utility>
#include <type_traits>

namespace detr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int A2_TILE = 2048;            // one 32-row x 64-byte head tile image
constexpr int A2_AUX = 256;              // one 64-lane x 4-byte piece (flag words / row statistics)
constexpr int A2_RING = 3;
constexpr float A2_SUM_LIMIT = 4096.0f;  // a lane's 16-key sum of exp2(score - reference) above this moves the reference (forward kernel)
// timing experiments only (scripts/experiments/attn2_ablate.sh; results are WRONG with any bit set): 1 = no DMA requests inside the key loop,
// 2 = no fragment reads / MFMAs / softmax inside the key loop, 4 = no softmax arithmetic (the scores go straight into the PV product)
#ifndef A2_ABLATE
#define A2_ABLATE 0
#endif
// (bit 8: the exponentials of the forward kernel as one v_add each -- what the transcendental rate costs)
__device__ __forceinline__ float a2_exp2(float x) { return (A2_ABLATE & 8) ? x + 1.0f : __builtin_amdgcn_exp2f(x); }

struct Attn2Args {
    const unsigned short *Q, *K, *V;
    unsigned short *O;
    float *LSE;
    const unsigned short *dO;
    unsigned short *dQ, *dK, *dV;
    float *stats;                        // [2][B*H*T]: delta / scale | -(lse * log2 e - log2 scale)
    const uint32_t *maskQ;               // keep bits, see attn2_dropmask_kernel
    int B, H, T, S;
    long long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;     // row strides in elements
    float qscale, drop_scale;
    uint32_t drop_thresh, drop_seed;
    const uint32_t *drop_step;
    int parts;                           // waves per workgroup = contiguous runs of the streamed dimension
    int xcd_map;                         // 1: workgroup id -> (row block, problem) keeps a (batch, head) problem on one XCD
};

#define A2_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)

__device__ __forceinline__ unsigned a2_pk(float a, float b) {
    bf16x2 r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ bf16x8 a2_pack8(const float *v) {
    u32x4 w;
    w[0] = a2_pk(v[0], v[1]); w[1] = a2_pk(v[2], v[3]); w[2] = a2_pk(v[4], v[5]); w[3] = a2_pk(v[6], v[7]);
    return __builtin_bit_cast(bf16x8, w);
}
// one 64-lane x 4-byte LDS-DMA piece (flag words, row statistics)
__device__ __forceinline__ void a2_dma4(u32x4 rs, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0");
}
// every LDS read of the wave has returned (a stage may be overwritten by a DMA request issued after this)
__device__ __forceinline__ void a2_lds_drained() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One streamed operand: 32-row tiles of 64-byte rows [row][32 bf16] at row stride ld, rows >= rows_end read as zeros
// (buffer range check; a request past the end moves no bytes).  Piece i = rows 16 i .. 16 i + 15: lane (r = lane >> 2, slot =
// lane & 3) requests source chunk slot ^ ((row >> 2) & 3) of its row, so chunk c of row r lands at byte r * 64 + (c ^ ((r >> 2) & 3)) * 16.
struct A2Src {
    u32x4 rs;
    unsigned v0, v1, tstride;
    __device__ __forceinline__ void init(const unsigned short *base, long long ld, int rows_end, int lane) {
        const long long bytes = rows_end > 0 ? ((long long)(rows_end - 1) * ld + 32) * 2 : 0;
        rs = ring_rsrc(base, (unsigned)bytes);
        const int r0 = lane >> 2, slot = lane & 3, r1 = r0 + 16;
        v0 = (unsigned)(r0 * ld * 2) + 16u * (unsigned)(slot ^ ((r0 >> 2) & 3));
        v1 = (unsigned)(r1 * ld * 2) + 16u * (unsigned)(slot ^ ((r1 >> 2) & 3));
        tstride = (unsigned)(32 * ld * 2);
    }
    __device__ __forceinline__ void issue(unsigned lds, int tile) const {
        const unsigned off = (unsigned)tile * tstride;
        ring_dma_piece(rs, lds, v0 + off);
        ring_dma_piece(rs, lds + 1024u, v1 + off);
    }
};

// per-lane byte offsets of the fragments inside a tile image
struct A2Frag {
    unsigned row[2];         // ds_read_b128: 8 consecutive columns 16 s + 8 hi .. of tile row (lane & 31)
    unsigned col[2][2];      // ds_read_b64_tr_b16: k-step s2, rows 16 s2 + 4 hi + {0..3} (j = 0) / + 8 (j = 1), of the lane's column
    __device__ __forceinline__ void init(int lane) {
        const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int s = 0; s < 2; ++s) row[s] = (unsigned)(l31 * 64 + (((2 * s + hi) ^ ((l31 >> 2) & 3)) * 16));
        const int g = lane >> 4, t = lane & 15;
        const int c = 2 * (g & 1) + ((t & 3) >> 1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = 16 * s2 + 4 * (g >> 1) + (t >> 2) + 8 * j;
                col[s2][j] = (unsigned)(r * 64 + ((c ^ ((r >> 2) & 3)) * 16) + 8 * (t & 1));
            }
    }
};
__device__ __forceinline__ bf16x8 a2_frag_row(const char *tile, const A2Frag &f, int s) {
    return *reinterpret_cast<const bf16x8 *>(tile + f.row[s]);
}
__device__ __forceinline__ bf16x8 a2_frag_col(const char *tile, const A2Frag &f, int s2) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(tile + f.col[s2][0]));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(tile + f.col[s2][1]));
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// 8 consecutive bf16 of one row from global memory (prologue operands); rows outside read as zeros
__device__ __forceinline__ bf16x8 a2_ld_row8(const unsigned short *base, long long ld, int row, int nrows, int col) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < nrows) v = *reinterpret_cast<const uint4 *>(base + (long long)row * ld + col);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ float a2_bf(unsigned short x) { return __builtin_bit_cast(float, (unsigned)x << 16); }
// 4 consecutive bf16 stores (one 8-byte request)
__device__ __forceinline__ void a2_st4(unsigned short *p, float a, float b, float c, float d) {
    *reinterpret_cast<uint2 *>(p) = make_uint2(a2_pk(a, b), a2_pk(c, d));
}
// Keep flag of register R as an all-ones / zero word: bit (R & 3) + 8 (R >> 2) of the flag word already shifted by 4 hi, ANDed onto the
// value.  v_bfe_i32 through inline assembly on purpose: when the compiler can see that the word is 0 / -1 it turns the AND into
// v_and (bit test) + v_cmp + v_cndmask and moves it behind the bf16 conversion (one v_cvt_pk per ELEMENT + v_perm): 4.5 instructions per
// element instead of 2.
template <int R>
__device__ __forceinline__ float a2_keep(float x, unsigned w) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w), "n"((R & 3) + 8 * (R >> 2)));
    return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & m);
}
// the same with the bit index in a register (the dK / dV kernel: bit = the lane's key)
__device__ __forceinline__ float a2_keep_bit(float x, unsigned w, int bit) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w), "v"(bit));
    return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & m);
}
template <int... R>
__device__ __forceinline__ void a2_keep16(float (&p)[16], unsigned w, std::integer_sequence<int, R...>) {
    ((p[R] = a2_keep<R>(p[R], w)), ...);
}
__device__ __forceinline__ void a2_keep16(float (&p)[16], unsigned w) { a2_keep16(p, w, std::make_integer_sequence<int, 16>{}); }
template <int S>
using a2_ic = std::integral_constant<int, S>;

// workgroup id -> (row block x, problem bh).  Blocks are dispatched round-robin over the 8 XCDs: with xcd_map the blocks of one
// (batch, head) problem share an XCD (its K / V stay in that XCD's L2), otherwise plain row-major order.
__device__ __forceinline__ void a2_block(const Attn2Args &a, int nblk, int &x, int &bh) {
    const int id = blockIdx.x;
    if (a.xcd_map) {
        const int xcd = id & 7, n = id >> 3;
        x = n % nblk;
        bh = (n / nblk) * 8 + xcd;
    } else {
        x = id % nblk;
        bh = id / nblk;
    }
}

extern __shared__ __attribute__((aligned(16))) char a2_smem[];

// The tile loops below walk the ring with the stage index as a COMPILE-TIME constant (three copies of the tile body per loop trip),
// so that every fragment read is `lane offset + immediate` and the stage bookkeeping costs no vector instruction.
// (an empty volatile asm inside a rarely taken, wave-uniform branch: the compiler must keep the branch -- if-converted, the ragged-tile
//  fix-ups would cost two vector instructions per score in EVERY tile)
#define A2_NO_IFCVT() asm volatile("" ::: "memory")
#define A2_TILE_LOOP(nt, body)                                   \
    for (int i_ = 0; i_ < (nt); i_ += 3) {                       \
        body(a2_ic<0>{}, i_);                                    \
        if (i_ + 1 < (nt)) body(a2_ic<1>{}, i_ + 1);             \
        if (i_ + 2 < (nt)) body(a2_ic<2>{}, i_ + 2);             \
    }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn2_fwd_kernel(Attn2Args a) {
    constexpr int NP = 4 + (DROP ? 1 : 0);                   // DMA requests per tile and wave
    constexpr int STAGE = 2 * A2_TILE + (DROP ? A2_AUX : 0);
    const int lane = threadIdx.x & 63, kp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int xb, bh;
    a2_block(a, (a.T + 63) / 64, xb, bh);
    const int b = bh / a.H, h = bh % a.H;
    const int q0 = xb * 64;
    const int ntiles = (a.S + 31) / 32, nth = (ntiles + a.parts - 1) / a.parts;
    const int t0 = kp * nth, t1 = min(ntiles, t0 + nth);
    const int nt = max(t1 - t0, 0);
    char *ring = a2_smem + kp * (A2_RING * STAGE);
    const unsigned ring_lds = ring_lds_addr(ring);

    A2Src ksrc, vsrc;
    const int rows_end = min(a.S, t1 * 32);
    ksrc.init(a.K + (long long)b * a.S * a.ldk + h * 32, a.ldk, rows_end, lane);
    vsrc.init(a.V + (long long)b * a.S * a.ldv + h * 32, a.ldv, rows_end, lane);
    // flag words: [bh][key tile][Tp] with Tp = 32 * ceil(T / 32); lane -> query q0 + lane (chain = lane >> 5)
    const int Tp = ((a.T + 31) / 32) * 32;
    u32x4 mrs = ring_rsrc(nullptr, 0);
    unsigned mvoff = 0;
    if constexpr (DROP) {
        // (queries past Tp: the offset of the last word -- the request stays inside the descriptor, its bits are never used)
        mrs = ring_rsrc(a.maskQ + (long long)bh * ntiles * Tp, (unsigned)((long long)t1 * Tp * 4));
        mvoff = (unsigned)min(q0 + lane, Tp - 1) * 4u;
    }
    auto issue = [&](int stg, int tile) {            // tile -> ring stage stg
        const unsigned st = ring_lds + (unsigned)(stg * STAGE);
        ksrc.issue(st, tile);
        vsrc.issue(st + A2_TILE, tile);
        if constexpr (DROP) a2_dma4(mrs, st + 2 * A2_TILE, mvoff + (unsigned)tile * (unsigned)Tp * 4u);
    };
    issue(0, t0);
    issue(1, t0 + 1);

    A2Frag fr;
    fr.init(lane);
    const unsigned short *Qb = a.Q + (long long)b * a.T * a.ldq + h * 32;
    bf16x8 qb[2][2];                     // [chain][k-step]: B operand of K Q^T, Q[q][16 s + 8 hi ..] (already in log2 units)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) qb[c][s] = a2_ld_row8(Qb, a.ldq, q0 + 32 * c + l31, a.T, 16 * s + 8 * hi);

    const float lg2scale = DROP ? __log2f(a.drop_scale) : 0.0f;
    f32x16 o[2], negm[2];
    float lsum[2] = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[c][r] = 0.0f; negm[c][r] = lg2scale; }

    auto tile = [&](auto stg_c, int i) {
        constexpr int STG = decltype(stg_c)::value;
        if constexpr ((A2_ABLATE & 1) == 0) {
            issue((STG + 2) % A2_RING, t0 + i + 2);     // (the stage of tile i - 1: its reads were drained at the end of that tile)
            ring_wait_vmcnt<2 * NP>();                  // tile i has landed
        }
        if constexpr ((A2_ABLATE & 2) != 0) return;
        const char *st = ring + STG * STAGE;
        const int kbase = (t0 + i) * 32;
        const bool first = (STG == 0) && i == 0;        // wave-uniform
        const bf16x8 k0 = a2_frag_row(st, fr, 0), k1 = a2_frag_row(st, fr, 1);
        f32x16 s[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            s[c] = A2_MFMA(k0, qb[c][0], negm[c]);
            s[c] = A2_MFMA(k1, qb[c][1], s[c]);
        }
        const bf16x8 v0 = a2_frag_col(st + A2_TILE, fr, 0), v1 = a2_frag_col(st + A2_TILE, fr, 1);
        unsigned w[2] = {0u, 0u};
        if constexpr (DROP) {
#pragma unroll
            for (int c = 0; c < 2; ++c) w[c] = *reinterpret_cast<const unsigned *>(st + 2 * A2_TILE + 128 * c + 4 * l31) >> (4 * hi);
        }
        const bool ragged = kbase + 32 > a.S;           // wave-uniform: the last tile only
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (ragged) {
                A2_NO_IFCVT();
                const int nv = a.S - kbase - 4 * hi;    // register r holds key kbase + 4 hi + (r & 3) + 8 (r >> 2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) >= nv) s[c][r] = -INFINITY;
            }
            if constexpr ((A2_ABLATE & 4) != 0) {
                float pa[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pa[r] = s[c][r];
                o[c] = A2_MFMA(v0, a2_pack8(pa), o[c]);
                o[c] = A2_MFMA(v1, a2_pack8(pa + 8), o[c]);
                continue;
            }
            if (first) {
                // a run's first tile sets the reference exactly to its maximum (o = l = 0 there)
                A2_NO_IFCVT();
                const float d = halves_max(tree_max16(s[c])) - lg2scale;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    negm[c][r] -= d;
                    s[c][r] -= d;
                }
            }
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = a2_exp2(s[c][r]);
            float ls = tree_sum16(p);
            // No maximum per tile: the lane's 16-key sum bounds every exponential in it.  Only when a sum leaves the safe range
            // (or is inf / NaN) -- a score far above everything seen so far -- the reference moves up and the tile is redone
            if (__builtin_amdgcn_ballot_w64(!(ls <= A2_SUM_LIMIT)) != 0ull) {
                A2_NO_IFCVT();
                // (the scores are recomputed from the stage rather than kept alive across the exponentials: 16 registers in the hot path)
                f32x16 s2 = A2_MFMA(a2_frag_row(st, fr, 0), qb[c][0], negm[c]);
                s2 = A2_MFMA(a2_frag_row(st, fr, 1), qb[c][1], s2);
                if (ragged) {
                    const int nv = a.S - kbase - 4 * hi;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if ((r & 3) + 8 * (r >> 2) >= nv) s2[r] = -INFINITY;
                }
                const float d = fmaxf(halves_max(tree_max16(s2)) - lg2scale, 0.0f);
                const float corr = a2_exp2(-d);
                lsum[c] *= corr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o[c][r] *= corr;
                    negm[c][r] -= d;
                    p[r] = a2_exp2(s2[r] - d);
                }
                ls = tree_sum16(p);
            }
            lsum[c] += ls;
            if constexpr (DROP) a2_keep16(p, w[c]);
            o[c] = A2_MFMA(v0, a2_pack8(p), o[c]);
            o[c] = A2_MFMA(v1, a2_pack8(p + 8), o[c]);
        }
        a2_lds_drained();
    };
    A2_TILE_LOOP(nt, tile)
    ring_wait_vmcnt<0>();                                // the trailing (empty) requests still write zeros into the ring
    float m[2], l[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        l[c] = halves_sum(lsum[c]);
        m[c] = (nt > 0) ? lg2scale - negm[c][0] : -INFINITY;
    }
    if (a.parts > 1) {
        // merge the partial softmaxes: (m, l, o) of runs 1 .. parts-1 travel through their own ring memory (component-major:
        // conflict-free), run 0 rescales everything to the common reference, in run order
        __syncthreads();
        if (kp > 0) {
            float *cb = reinterpret_cast<float *>(ring) + lane;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                cb[(18 * c) * 64] = m[c];
                cb[(18 * c + 1) * 64] = l[c];
#pragma unroll
                for (int r = 0; r < 16; ++r) cb[(18 * c + 2 + r) * 64] = o[c][r];
            }
        }
        __syncthreads();
        if (kp > 0) return;
        for (int k = 1; k < a.parts; ++k) {
            const float *cb = reinterpret_cast<const float *>(a2_smem + k * (A2_RING * STAGE)) + lane;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float mb = cb[(18 * c) * 64], lb = cb[(18 * c + 1) * 64];
                const float mn = fmaxf(m[c], mb);
                const float ca = a2_exp2(m[c] - mn), cbf = a2_exp2(mb - mn);       // (a run without keys: m = -inf, l = 0)
                l[c] = l[c] * ca + lb * cbf;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[c][r] = o[c][r] * ca + cb[(18 * c + 2 + r) * 64] * cbf;
                m[c] = mn;
            }
        }
    }
    const float dsc = DROP ? a.drop_scale : 1.0f;         // l carries the dropout scale (it rides in the exponent)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int tq = q0 + 32 * c + l31;
        if (tq < a.T) {
            const float inv = dsc / l[c];
            unsigned short *Ob = a.O + ((long long)b * a.T + tq) * a.ldo + h * 32 + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                a2_st4(Ob + 8 * g, o[c][4 * g] * inv, o[c][4 * g + 1] * inv, o[c][4 * g + 2] * inv, o[c][4 * g + 3] * inv);
            if (hi == 0) a.LSE[(long long)bh * a.T + tq] = m[c] * AT_LN2 + logf(l[c] / dsc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward 1/2: dQ per query block (streams the keys) + the row statistics for the dK / dV kernel
// ------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn2_bwd_dq_kernel(Attn2Args a) {
    constexpr int NP = 4 + (DROP ? 1 : 0);
    constexpr int STAGE = 2 * A2_TILE + (DROP ? A2_AUX : 0);
    const int lane = threadIdx.x & 63, kp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int xb, bh;
    a2_block(a, (a.T + 63) / 64, xb, bh);
    const int b = bh / a.H, h = bh % a.H;
    const int q0 = xb * 64;
    const int ntiles = (a.S + 31) / 32, nth = (ntiles + a.parts - 1) / a.parts;
    const int t0 = kp * nth, t1 = min(ntiles, t0 + nth);
    const int nt = max(t1 - t0, 0);
    char *ring = a2_smem + kp * (A2_RING * STAGE);
    const unsigned ring_lds = ring_lds_addr(ring);

    A2Src ksrc, vsrc;
    const int rows_end = min(a.S, t1 * 32);
    ksrc.init(a.K + (long long)b * a.S * a.ldk + h * 32, a.ldk, rows_end, lane);
    vsrc.init(a.V + (long long)b * a.S * a.ldv + h * 32, a.ldv, rows_end, lane);
    const int Tp = ((a.T + 31) / 32) * 32;
    u32x4 mrs = ring_rsrc(nullptr, 0);
    unsigned mvoff = 0;
    if constexpr (DROP) {
        mrs = ring_rsrc(a.maskQ + (long long)bh * ntiles * Tp, (unsigned)((long long)t1 * Tp * 4));
        mvoff = (unsigned)min(q0 + lane, Tp - 1) * 4u;
    }
    auto issue = [&](int stg, int tile) {
        const unsigned st = ring_lds + (unsigned)(stg * STAGE);
        ksrc.issue(st, tile);
        vsrc.issue(st + A2_TILE, tile);
        if constexpr (DROP) a2_dma4(mrs, st + 2 * A2_TILE, mvoff + (unsigned)tile * (unsigned)Tp * 4u);
    };
    issue(0, t0);
    issue(1, t0 + 1);

    A2Frag fr;
    fr.init(lane);
    const unsigned short *Qb = a.Q + (long long)b * a.T * a.ldq + h * 32;
    const unsigned short *Db = a.dO + (long long)b * a.T * a.lddo + h * 32;
    const unsigned short *Ob = a.O + (long long)b * a.T * a.ldo + h * 32;
    const float lg2scale = DROP ? __log2f(a.drop_scale) : 0.0f;
    const float inv_scale = DROP ? 1.0f / a.drop_scale : 1.0f;
    bf16x8 qb[2][2], dob[2][2];
    float Lq[2], dlq[2];                 // lse * log2 e - log2 scale (exp2(s - Lq) = P * scale), delta / scale
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int tq = q0 + 32 * c + l31;
        float dl = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qb[c][s] = a2_ld_row8(Qb, a.ldq, tq, a.T, 16 * s + 8 * hi);
            dob[c][s] = a2_ld_row8(Db, a.lddo, tq, a.T, 16 * s + 8 * hi);
            const bf16x8 ov = a2_ld_row8(Ob, a.ldo, tq, a.T, 16 * s + 8 * hi);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += (float)dob[c][s][j] * (float)ov[j];
        }
        dl = halves_sum(dl);
        const bool ok = tq < a.T;
        Lq[c] = ok ? a.LSE[(long long)bh * a.T + tq] * AT_LOG2E - lg2scale : INFINITY;      // padded queries: P = 0
        dlq[c] = dl * inv_scale;
        if (ok && hi == 0 && kp == 0) {
            a.stats[(long long)bh * a.T + tq] = dlq[c];
            a.stats[(long long)a.B * a.H * a.T + (long long)bh * a.T + tq] = -Lq[c];       // (negated: the dK / dV kernel feeds it to its MFMAs as C)
        }
    }
    f32x16 dq[2], negL[2];               // -L' is the C operand of the first score MFMA: exp2(s - L') costs no subtraction
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq[c][r] = 0.0f; negL[c][r] = -Lq[c]; }

    auto tile = [&](auto stg_c, int i) {
        constexpr int STG = decltype(stg_c)::value;
        issue((STG + 2) % A2_RING, t0 + i + 2);
        ring_wait_vmcnt<2 * NP>();
        const char *st = ring + STG * STAGE;
        const int kbase = (t0 + i) * 32;
        const bf16x8 k0 = a2_frag_row(st, fr, 0), k1 = a2_frag_row(st, fr, 1);
        const bf16x8 vr0 = a2_frag_row(st + A2_TILE, fr, 0), vr1 = a2_frag_row(st + A2_TILE, fr, 1);
        f32x16 s[2], dp[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[c][r] = 0.0f;
            s[c] = A2_MFMA(k0, qb[c][0], negL[c]);
            dp[c] = A2_MFMA(vr0, dob[c][0], dp[c]);
            s[c] = A2_MFMA(k1, qb[c][1], s[c]);
            dp[c] = A2_MFMA(vr1, dob[c][1], dp[c]);
        }
        const bf16x8 kc0 = a2_frag_col(st, fr, 0), kc1 = a2_frag_col(st, fr, 1);
        unsigned w[2] = {0u, 0u};
        if constexpr (DROP) {
#pragma unroll
            for (int c = 0; c < 2; ++c) w[c] = *reinterpret_cast<const unsigned *>(st + 2 * A2_TILE + 128 * c + 4 * l31) >> (4 * hi);
        }
        const bool ragged = kbase + 32 > a.S;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float ds[16], dpm[16];
            // dS = P (drop(dP) - delta) = (P scale) ((keep ? dP : 0) - delta / scale)
#pragma unroll
            for (int r = 0; r < 16; ++r) dpm[r] = dp[c][r];
            if constexpr (DROP) a2_keep16(dpm, w[c]);
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = fast_exp2(s[c][r]) * (dpm[r] - dlq[c]);
            if (ragged) {                               // (zero K rows make s = 0: exp2(-Lq) may be anything)
                A2_NO_IFCVT();
                const int nv = a.S - kbase - 4 * hi;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) >= nv) ds[r] = 0.0f;
            }
            dq[c] = A2_MFMA(kc0, a2_pack8(ds), dq[c]);
            dq[c] = A2_MFMA(kc1, a2_pack8(ds + 8), dq[c]);
        }
        a2_lds_drained();
    };
    A2_TILE_LOOP(nt, tile)
    ring_wait_vmcnt<0>();
    if (a.parts > 1) {
        __syncthreads();
        if (kp > 0) {
            float *cb = reinterpret_cast<float *>(ring) + lane;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) cb[(16 * c + r) * 64] = dq[c][r];
        }
        __syncthreads();
        if (kp > 0) return;
        for (int k = 1; k < a.parts; ++k) {
            const float *cb = reinterpret_cast<const float *>(a2_smem + k * (A2_RING * STAGE)) + lane;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) dq[c][r] += cb[(16 * c + r) * 64];
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int tq = q0 + 32 * c + l31;
        if (tq < a.T) {
            unsigned short *dst = a.dQ + ((long long)b * a.T + tq) * a.lddq + h * 32 + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g)      // gradient w.r.t. the UNSCALED q
                a2_st4(dst + 8 * g, dq[c][4 * g] * a.qscale, dq[c][4 * g + 1] * a.qscale, dq[c][4 * g + 2] * a.qscale, dq[c][4 * g + 3] * a.qscale);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward 2/2: dK, dV per key block (streams the queries): lane l holds key (l & 31) of a chain and 16 queries krow(r, hi)
// ------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn2_bwd_dkv_kernel(Attn2Args a) {
    constexpr int NP = 5 + (DROP ? 1 : 0);
    constexpr int STAGE = 2 * A2_TILE + A2_AUX + (DROP ? A2_AUX : 0);
    const int lane = threadIdx.x & 63, qp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int xb, bh;
    a2_block(a, (a.S + 63) / 64, xb, bh);
    const int b = bh / a.H, h = bh % a.H;
    const int key0 = xb * 64;
    const int ntiles = (a.T + 31) / 32, nth = (ntiles + a.parts - 1) / a.parts;
    const int t0 = qp * nth, t1 = min(ntiles, t0 + nth);
    const int nt = max(t1 - t0, 0);
    char *ring = a2_smem + qp * (A2_RING * STAGE);
    const unsigned ring_lds = ring_lds_addr(ring);

    A2Src qsrc, dsrc;
    const int rows_end = min(a.T, t1 * 32);
    qsrc.init(a.Q + (long long)b * a.T * a.ldq + h * 32, a.ldq, rows_end, lane);
    dsrc.init(a.dO + (long long)b * a.T * a.lddo + h * 32, a.lddo, rows_end, lane);
    // row statistics of a query tile: lanes 0..31 <- lse' [q], lanes 32..63 <- delta' [q]  (query q = tile * 32 + l31; a tile's last
    // queries may lie past T: they read a neighbour's numbers -- finite, and zeroed below)
    const long long BHT = (long long)a.B * a.H * a.T;
    const u32x4 srs = ring_rsrc(a.stats, (unsigned)(2 * BHT * 4));
    const unsigned svoff = (unsigned)(((hi ? 0ll : BHT) + (long long)bh * a.T + l31) * 4);
    // flag words [bh][key tile][Tp] (bit = key): lanes 0..31 <- the words of (key tile of chain 0, queries tile * 32 + l31), lanes 32..63 <- chain 1
    const int Tp = ntiles * 32, nkt = (a.S + 31) / 32;
    u32x4 mrs = ring_rsrc(nullptr, 0);
    unsigned mvoff = 0;
    if constexpr (DROP) {
        mrs = ring_rsrc(a.maskQ + (long long)bh * nkt * Tp, (unsigned)((long long)nkt * Tp * 4));
        const int kt = min(key0 / 32 + hi, nkt - 1);            // (a second chain past the last key tile: any words, its keys are not stored)
        mvoff = (unsigned)(kt * Tp + l31) * 4u;
    }
    auto issue = [&](int stg, int tile) {
        const unsigned st = ring_lds + (unsigned)(stg * STAGE);
        qsrc.issue(st, tile);
        dsrc.issue(st + A2_TILE, tile);
        a2_dma4(srs, st + 2 * A2_TILE, (tile < t1) ? svoff + (unsigned)tile * 128u : BUF_OOB);
        if constexpr (DROP) a2_dma4(mrs, st + 2 * A2_TILE + A2_AUX, (tile < t1) ? mvoff + (unsigned)tile * 128u : BUF_OOB);
    };
    issue(0, t0);
    issue(1, t0 + 1);

    A2Frag fr;
    fr.init(lane);
    const unsigned short *Kb = a.K + (long long)b * a.S * a.ldk + h * 32;
    const unsigned short *Vb = a.V + (long long)b * a.S * a.ldv + h * 32;
    bf16x8 kb[2][2], vb[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            kb[c][s] = a2_ld_row8(Kb, a.ldk, key0 + 32 * c + l31, a.S, 16 * s + 8 * hi);
            vb[c][s] = a2_ld_row8(Vb, a.ldv, key0 + 32 * c + l31, a.S, 16 * s + 8 * hi);
        }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[c][r] = 0.0f; dv[c][r] = 0.0f; }

    auto tile = [&](auto stg_c, int i) {
        constexpr int STG = decltype(stg_c)::value;
        issue((STG + 2) % A2_RING, t0 + i + 2);
        ring_wait_vmcnt<2 * NP>();
        const char *st = ring + STG * STAGE;
        const int qbase = (t0 + i) * 32;
        const bf16x8 qr0 = a2_frag_row(st, fr, 0), qr1 = a2_frag_row(st, fr, 1);
        const bf16x8 dr0 = a2_frag_row(st + A2_TILE, fr, 0), dr1 = a2_frag_row(st + A2_TILE, fr, 1);
        const bf16x8 qc0 = a2_frag_col(st, fr, 0), qc1 = a2_frag_col(st, fr, 1);
        const bf16x8 dc0 = a2_frag_col(st + A2_TILE, fr, 0), dc1 = a2_frag_col(st + A2_TILE, fr, 1);
        // lse' / delta' of the 16 queries krow(r, hi): four runs of four consecutive floats each
        f32x16 negLr;                   // -lse' of the lane's 16 queries = the C operand of the score MFMAs (register r <-> query krow(r, hi))
        float Dr[16];
        const float *sl = reinterpret_cast<const float *>(st + 2 * A2_TILE);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 x = *reinterpret_cast<const float4 *>(sl + 8 * g + 4 * hi);
            const float4 y = *reinterpret_cast<const float4 *>(sl + 32 + 8 * g + 4 * hi);
            negLr[4 * g] = x.x; negLr[4 * g + 1] = x.y; negLr[4 * g + 2] = x.z; negLr[4 * g + 3] = x.w;
            Dr[4 * g] = y.x; Dr[4 * g + 1] = y.y; Dr[4 * g + 2] = y.z; Dr[4 * g + 3] = y.w;
        }
        const bool ragged = qbase + 32 > a.T;           // wave-uniform: the last query tile only
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.0f;
            s = A2_MFMA(qr0, kb[c][0], negLr);
            dp = A2_MFMA(dr0, vb[c][0], dp);
            s = A2_MFMA(qr1, kb[c][1], s);
            dp = A2_MFMA(dr1, vb[c][1], dp);
            float p[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = fast_exp2(s[r]);                                         // P (* scale with dropout)
                ds[r] = -(p[r] * Dr[r]);
            }
            if constexpr (DROP) {                                               // dV uses the dropped probabilities
                // the flag of (query krow(r, hi), this lane's key) is bit l31 of that query's word: one broadcast read per four queries
                const unsigned *mw = reinterpret_cast<const unsigned *>(st + 2 * A2_TILE + A2_AUX + 128 * c);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint4 wq = *reinterpret_cast<const uint4 *>(mw + 8 * g + 4 * hi);
                    p[4 * g] = a2_keep_bit(p[4 * g], wq.x, l31);
                    p[4 * g + 1] = a2_keep_bit(p[4 * g + 1], wq.y, l31);
                    p[4 * g + 2] = a2_keep_bit(p[4 * g + 2], wq.z, l31);
                    p[4 * g + 3] = a2_keep_bit(p[4 * g + 3], wq.w, l31);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = __builtin_fmaf(p[r], dp[r], ds[r]);   // (P scale) ((keep ? dP : 0) - delta / scale)
            if (ragged) {                               // (zero Q / dO rows and a neighbour's statistics: exp2 may be anything)
                A2_NO_IFCVT();
                const int nv = a.T - qbase - 4 * hi;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) >= nv) { p[r] = 0.0f; ds[r] = 0.0f; }
            }
            dv[c] = A2_MFMA(dc0, a2_pack8(p), dv[c]);
            dk[c] = A2_MFMA(qc0, a2_pack8(ds), dk[c]);
            dv[c] = A2_MFMA(dc1, a2_pack8(p + 8), dv[c]);
            dk[c] = A2_MFMA(qc1, a2_pack8(ds + 8), dk[c]);
        }
        a2_lds_drained();
    };
    A2_TILE_LOOP(nt, tile)
    ring_wait_vmcnt<0>();
    if (a.parts > 1) {
        // partial sums of runs 1 .. parts-1 through LDS, one chain at a time (a run's ring holds 2 x 16 x 64 floats), fixed order
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            __syncthreads();
            if (qp > 0) {
                float *cb = reinterpret_cast<float *>(ring) + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) { cb[r * 64] = dk[c][r]; cb[(16 + r) * 64] = dv[c][r]; }
            }
            __syncthreads();
            if (qp == 0) {
                for (int k = 1; k < a.parts; ++k) {
                    const float *cb = reinterpret_cast<const float *>(a2_smem + k * (A2_RING * STAGE)) + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { dk[c][r] += cb[r * 64]; dv[c][r] += cb[(16 + r) * 64]; }
                }
            }
        }
        if (qp > 0) return;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int sk = key0 + 32 * c + l31;
        if (sk < a.S) {
            unsigned short *dkp = a.dK + ((long long)b * a.S + sk) * a.lddk + h * 32 + 4 * hi;
            unsigned short *dvp = a.dV + ((long long)b * a.S + sk) * a.lddv + h * 32 + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g) {      // (the stored Q carries scale * log2 e: dK = ln 2 * dS^T Q)
                a2_st4(dkp + 8 * g, dk[c][4 * g] * AT_LN2, dk[c][4 * g + 1] * AT_LN2, dk[c][4 * g + 2] * AT_LN2, dk[c][4 * g + 3] * AT_LN2);
                a2_st4(dvp + 8 * g, dv[c][4 * g], dv[c][4 * g + 1], dv[c][4 * g + 2], dv[c][4 * g + 3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dropout keep bits of one attention site and step (common.h::drop_keep of element (row * Sp + key), row = bh * T + query):
//   mask[(bh * nkt + kt) * Tp + q]      bit j = keep(q, kt * 32 + j)        Tp = 32 ceil(T / 32), nkt = ceil(S / 32)
// One lane per (query, key tile): 16 pair hashes give its word.  All three kernels read this layout (the forward / dQ kernels one
// word per lane and tile, the dK / dV kernel -- whose lanes own keys -- the words of its 16 queries with the lane's key as the bit index).
// ------------------------------------------------------------------------------------------------
constexpr int A2_MASK_SITES = 32;        // attention sites one keep-bit launch covers (a DETR step has 18)
struct A2MaskSite {
    uint32_t *mask;
    int T, S, nkt, Tp, xblocks;          // xblocks = ceil(Tp / 256)
    uint32_t site;
    long long block0;                    // first workgroup of this site in the launch
};
struct A2MaskArgs {
    A2MaskSite s[A2_MASK_SITES];
    int n, BH;
    uint32_t thresh16;
    const uint32_t *step;
};
__global__ __launch_bounds__(256) void attn2_dropmask_kernel(A2MaskArgs a) {
    int si = 0;
#pragma unroll 1
    for (int i = 1; i < a.n; ++i)
        if ((long long)blockIdx.x >= a.s[i].block0) si = i;
    const A2MaskSite &m = a.s[si];
    const long long lb = (long long)blockIdx.x - m.block0;               // (x block, key tile, problem) of this site
    const int xb = (int)(lb % m.xblocks);
    const int kt = (int)((lb / m.xblocks) % m.nkt), bh = (int)(lb / ((long long)m.xblocks * m.nkt));
    const int q = xb * 256 + threadIdx.x;
    if (q >= m.Tp) return;
    const uint32_t key = drop_key(m.site, a.step);
    const unsigned long long Sp = (unsigned long long)((m.S + 1) & ~1);
    const unsigned long long pb = (((unsigned long long)bh * m.T + q) * Sp + (unsigned)(kt * 32)) >> 1;
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t hsh = drop_hash(key, pb + i);
        w |= ((hsh & 0xFFFFu) >= a.thresh16 ? 1u : 0u) << (2 * i);
        w |= ((hsh >> 16) >= a.thresh16 ? 1u : 0u) << (2 * i + 1);
    }
    m.mask[((long long)bh * m.nkt + kt) * m.Tp + q] = w;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int a2_stage_bytes(int kind, bool drop) {       // kind 0: forward / dQ, 1: dK / dV
    return 2 * A2_TILE + (kind ? A2_AUX : 0) + (drop ? A2_AUX : 0);
}

// waves per workgroup (runs of the streamed dimension).  Measured (scripts/micro_attn2.py, profiles/r06_micro_attn2.txt): four runs are
// best or within 3 % of the best on every shape of the step (encoder 1050 x 1050: forward 49 / 32 / 30 / 30 / 48 us at 5 / 2 / 3 / 4 / 6 runs --
// from five runs on, a workgroup's rings leave room for two workgroups per CU only); short streams keep at least 4 tiles per run.
// DETR_HIP_ATTN_SPLIT=n forces.
static int a2_parts(int blocks, int tiles) {
    const int force = tune(T_ATTN_SPLIT);
    int p = (force >= 1 && force <= 8) ? force : 4;
    const int cap = tiles >= 4 ? tiles / 4 : 1;
    if (p > cap) p = cap;
    return p < 1 ? 1 : p;
}

template <typename K>
static int a2_launch(K kernel, const Attn2Args &a, int blocks, int lds, hipStream_t s, const char *what) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        DETR_REQUIRE(e == hipSuccess, "%s: cannot reserve %d bytes of LDS: %s", what, lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3((unsigned)(64 * a.parts)), (size_t)lds, s, a);
    return 0;
}

static void a2_grid(Attn2Args &a, int rows, int &nblk, int &blocks) {
    const int bh = a.B * a.H;
    nblk = cdiv(rows, 64);
    blocks = nblk * bh;
    a.xcd_map = (bh % 8 == 0) ? 1 : 0;
}

int attn2_fwd_launch(Attn2Args a, hipStream_t s) {
    int nblk, blocks;
    a2_grid(a, a.T, nblk, blocks);
    const bool drop = a.drop_scale != 0.0f;
    a.parts = a2_parts(blocks, cdiv(a.S, 32));
    const int lds = a.parts * A2_RING * a2_stage_bytes(0, drop);
    if (drop) { if (a2_launch(attn2_fwd_kernel<true>, a, blocks, lds, s, "attention fwd (bf16 io)")) return -1; }
    else if (a2_launch(attn2_fwd_kernel<false>, a, blocks, lds, s, "attention fwd (bf16 io)")) return -1;
    DETR_LAUNCH_CHECK("attention fwd (bf16 io)");
    return 0;
}

int attn2_bwd_launch(Attn2Args a, hipStream_t s) {
    int nblk, blocks;
    const bool drop = a.drop_scale != 0.0f;
    a2_grid(a, a.T, nblk, blocks);
    a.parts = a2_parts(blocks, cdiv(a.S, 32));
    int lds = a.parts * A2_RING * a2_stage_bytes(0, drop);
    if (drop) { if (a2_launch(attn2_bwd_dq_kernel<true>, a, blocks, lds, s, "attention bwd dq (bf16 io)")) return -1; }
    else if (a2_launch(attn2_bwd_dq_kernel<false>, a, blocks, lds, s, "attention bwd dq (bf16 io)")) return -1;
    DETR_LAUNCH_CHECK("attention bwd dq (bf16 io)");
    a2_grid(a, a.S, nblk, blocks);
    a.parts = a2_parts(blocks, cdiv(a.T, 32));
    lds = a.parts * A2_RING * a2_stage_bytes(1, drop);
    if (drop) { if (a2_launch(attn2_bwd_dkv_kernel<true>, a, blocks, lds, s, "attention bwd dkv (bf16 io)")) return -1; }
    else if (a2_launch(attn2_bwd_dkv_kernel<false>, a, blocks, lds, s, "attention bwd dkv (bf16 io)")) return -1;
    DETR_LAUNCH_CHECK("attention bwd dkv (bf16 io)");
    return 0;
}

static long long a2_mask_words(int B, int H, int T, int S) {
    return 1ll * B * H * cdiv(T, 32) * cdiv(S, 32) * 32;
}

static int a2_check_ld(long long ld, int H, const char *what) {
    DETR_REQUIRE(ld >= (long long)H * 32 && ld % 8 == 0, "attention (bf16 io): row stride of %s (%lld) must be >= heads*32 and a multiple of 8", what, ld);
    return 0;
}

static int a2_from_desc(const detr_attn_desc *d, int bwd, Attn2Args &a) {
    DETR_REQUIRE(d->compute == 1, "attention: bf16 operands (io_dtype = 1) need compute = 1");
    DETR_REQUIRE(d->q && d->k && d->v && d->o && d->lse, "attention: null operand");
    DETR_REQUIRE(d->B > 0 && d->H > 0 && d->T > 0 && d->S > 0, "attention: bad shape B=%d H=%d T=%d S=%d", d->B, d->H, d->T, d->S);
    if (a2_check_ld(d->ldq, d->H, "q") || a2_check_ld(d->ldk, d->H, "k") || a2_check_ld(d->ldv, d->H, "v") || a2_check_ld(d->ldo, d->H, "o")) return -1;
    DETR_REQUIRE(aligned16(d->q) && aligned16(d->k) && aligned16(d->v) && aligned16(d->o), "attention: operands must be 16-byte aligned");
    DETR_REQUIRE(d->dropout_p >= 0.0f && d->dropout_p < 1.0f, "attention: dropout p=%f out of range", d->dropout_p);
    DETR_REQUIRE(d->scale > 0.0f, "attention: scale must be positive");
    const long long span_q = ((long long)d->T + 96) * (d->ldq > d->ldo ? d->ldq : d->ldo) * 2;
    const long long span_k = ((long long)d->S + 96) * (d->ldk > d->ldv ? d->ldk : d->ldv) * 2;
    DETR_REQUIRE(span_q < BUF_MAX_BYTES && span_k < BUF_MAX_BYTES, "attention (bf16 io): one batch entry of an operand spans more than 4 GB");
    DETR_REQUIRE(d->dropout_p == 0.0f || d->dropmask, "attention (bf16 io): dropout needs the keep bits (detr_hip_attention_dropmask)");
    DETR_REQUIRE(a2_mask_words(d->B, d->H, d->T, d->S) * 4 < BUF_MAX_BYTES, "attention (bf16 io): keep bits exceed 4 GB");
    a = Attn2Args{};
    a.Q = reinterpret_cast<const unsigned short *>(d->q); a.K = reinterpret_cast<const unsigned short *>(d->k);
    a.V = reinterpret_cast<const unsigned short *>(d->v); a.O = reinterpret_cast<unsigned short *>(d->o);
    a.LSE = d->lse;
    a.B = d->B; a.H = d->H; a.T = d->T; a.S = d->S;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo;
    a.qscale = d->scale;
    a.drop_scale = d->dropout_p > 0.0f ? 1.0f / (1.0f - d->dropout_p) : 0.0f;
    a.drop_thresh = drop_thresh16(d->dropout_p);
    a.drop_seed = d->dropout_site;
    a.drop_step = d->dropout_step;
    a.maskQ = d->dropmask;
    if (bwd) {
        DETR_REQUIRE(d->d_o && d->dq && d->dk && d->dv && d->delta, "attention bwd: null operand");
        if (a2_check_ld(d->ldd_o, d->H, "d_o") || a2_check_ld(d->lddq, d->H, "dq") || a2_check_ld(d->lddk, d->H, "dk") ||
            a2_check_ld(d->lddv, d->H, "dv")) return -1;
        DETR_REQUIRE(aligned16(d->d_o) && aligned16(d->dq) && aligned16(d->dk) && aligned16(d->dv), "attention bwd: gradients must be 16-byte aligned");
        DETR_REQUIRE(((long long)d->T + 96) * d->ldd_o * 2 < BUF_MAX_BYTES, "attention (bf16 io): d_o spans more than 4 GB");
        DETR_REQUIRE(2ll * d->B * d->H * d->T * 4 < BUF_MAX_BYTES, "attention (bf16 io): row statistics exceed 4 GB");
        a.dO = reinterpret_cast<const unsigned short *>(d->d_o);
        a.dQ = reinterpret_cast<unsigned short *>(d->dq); a.dK = reinterpret_cast<unsigned short *>(d->dk);
        a.dV = reinterpret_cast<unsigned short *>(d->dv);
        a.stats = d->delta;
        a.lddo = d->ldd_o; a.lddq = d->lddq; a.lddk = d->lddk; a.lddv = d->lddv;
    }
    return 0;
}

int attn2_fwd_from_desc(const detr_attn_desc *d, hipStream_t s) {
    Attn2Args a;
    if (a2_from_desc(d, 0, a)) return -1;
    return attn2_fwd_launch(a, s);
}
int attn2_bwd_from_desc(const detr_attn_desc *d, hipStream_t s) {
    Attn2Args a;
    if (a2_from_desc(d, 1, a)) return -1;
    return attn2_bwd_launch(a, s);
}

}  // namespace detr

extern "C" int64_t detr_hip_attention_dropmask_words(int32_t B, int32_t H, int32_t T, int32_t S) {
    if (B <= 0 || H <= 0 || T <= 0 || S <= 0) return -1;
    return detr::a2_mask_words(B, H, T, S);
}

extern "C" int detr_hip_attention_dropmask_many(const detr_attn_desc *d, int32_t n, void *stream) {
    using namespace detr;
    DETR_REQUIRE(d && n >= 1, "attention dropmask: null descriptors");
    int done = 0;
    while (done < n) {
        A2MaskArgs a;
        a.n = (n - done) < A2_MASK_SITES ? (n - done) : A2_MASK_SITES;
        long long blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const detr_attn_desc &e = d[done + i];
            DETR_REQUIRE(e.B > 0 && e.H > 0 && e.T > 0 && e.S > 0, "attention dropmask: bad shape B=%d H=%d T=%d S=%d", e.B, e.H, e.T, e.S);
            DETR_REQUIRE(e.dropmask, "attention dropmask: null output");
            DETR_REQUIRE(e.dropout_p > 0.0f && e.dropout_p < 1.0f, "attention dropmask: dropout p=%f out of range", e.dropout_p);
            DETR_REQUIRE(e.B * e.H == d[done].B * d[done].H && e.dropout_p == d[done].dropout_p && e.dropout_step == d[done].dropout_step,
                         "attention dropmask: the sites of one call share B*H, the dropout rate and the step seed");
            A2MaskSite &m = a.s[i];
            m.mask = e.dropmask; m.T = e.T; m.S = e.S; m.nkt = cdiv(e.S, 32); m.Tp = cdiv(e.T, 32) * 32; m.xblocks = cdiv(m.Tp, 256);
            m.site = e.dropout_site; m.block0 = blocks;
            blocks += (long long)m.xblocks * m.nkt * e.B * e.H;
        }
        for (int i = a.n; i < A2_MASK_SITES; ++i) a.s[i] = a.s[0];
        DETR_REQUIRE(blocks < 0x7FFFFFFFll, "attention dropmask: grid too large");
        a.BH = d[done].B * d[done].H;
        a.thresh16 = drop_thresh16(d[done].dropout_p);
        a.step = d[done].dropout_step;
        hipLaunchKernelGGL(attn2_dropmask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
        DETR_LAUNCH_CHECK("attention dropmask");
        done += a.n;
    }
    return 0;
}

extern "C" int detr_hip_attention_dropmask(const detr_attn_desc *d, void *stream) {
    DETR_REQUIRE(d, "attention dropmask: null descriptor");
    return detr_hip_attention_dropmask_many(d, 1, stream);
}
