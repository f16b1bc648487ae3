// gemm_x3.h -- "f32x3" tile GEMM (detr_gemm_desc.compute = 2): fp32 operands in HBM, fp32 ACCURACY, bf16 matrix pipe.
//
// Why: on gfx950 v_mfma_f32_32x32x2_f32 (the exact-fp32 parity mode's instruction) runs at 1/16 of the rate of
// v_mfma_f32_32x32x16_bf16.  An fp32 value is EXACTLY the sum of three bf16 values (gemm_core.h: split3_pair), so a product
// x*y is the sum of nine bf16 products; the six largest carry it to 2^-23 relative -- one fp32 rounding -- and each of them
// is exact in the fp32 accumulator's input.  6 x 32 matrix-pipe cycles per 32x32x16 block against 8 x 64 on the fp32 instruction.
//
// Round 6 built this twice.  First form (gemm_core.h: mma_ktile_split3, still used by the 3x3 convolutions and the stem): the
// exact kernel's fp32 LDS tile, every wave splits the fragments it reads -- each value is split by the two waves that share it
// and gathered with ds_read_b32: 1.0-1.25x the exact kernel (VALU and latency bound, profiles/r06_micro_split3.txt).  This
// file is the second form: the split happens ONCE per value, on the way from the global-load registers into LDS, which then
// holds three bf16 images per operand in the bf16 engine's layouts (gemm_bf16_core.h: [row][k] rows of 80 B read with
// ds_read_b128 for a K-contiguous operand, the transpose-read image + ds_read_b64_tr_b16 for an MN-contiguous one).
//   * 256 threads = 2 x 2 waves, tiles 128 x 128 (wave tile 64 x 64: 12 fragment reads feed 48 MFMAs per k-step) or 64 x 64.
//   * K tile 32.  ONE LDS buffer (3 x (BM + BN) x 80 B = 60 KB at 128 x 128: two workgroups per CU) and one register set:
//     loop = [split + store tile t from registers] [request tile t+1] barrier [2 k-steps x 6 x TM x TN MFMAs] barrier --
//     a workgroup's split phase (VALU, LDS writes) runs under the co-resident workgroup's MFMA phase.
//   * (measured and NOT kept, scripts/experiments/gemm_x3pp.h: one 8-wave workgroup whose two 4-wave groups are held in anti-phase by the
//     workgroup barrier -- one splits while the other multiplies.  Bit-identical, 1.3-1.6x SLOWER (M33600 N256 K1024: 154 -> 246 us): two
//     independent workgroups drift into whatever interleaving the SIMD arbitration finds; the barrier pins every half-iteration to the slower
//     of the two phases.  profiles/r06_ab_results.txt #7)
//   * epilogue, split-K slabs, fused row sums (from the fp32 registers, before any rounding), XCD remap: the tile engine's.
#pragma once
#include "gemm_kernels.h"

namespace detr {

constexpr int X3_BK = BF_BK;     // 32
constexpr int X3_LD = BF_LD;     // 40 bf16 per LDS row

template <int BM, int BN, int NBUF = 1>
struct X3Smem {
    unsigned short A[NBUF][3][BM][X3_LD];      // pieces h, m, l
    unsigned short B[NBUF][3][BN][X3_LD];
};
template <int BM, int BN, int NBUF = 1>
struct X3SmemBytes {
    static constexpr int TILES = (int)sizeof(X3Smem<BM, BN, NBUF>);
    static constexpr int STAGE = 4 * 32 * (BN / 2 + 4) * 4;
    static constexpr int VALUE = TILES > STAGE ? TILES : STAGE;
};

// ---- fp32 operand loaders in the per-tile descriptor form of the bf16 engine (gemm_bf16_core.h: LoaderKh / LoaderMNth::load_tile): the K
// advance lives in the buffer DESCRIPTOR (base += k0, num_records shrinks: scalar instructions), the per-lane offsets are loop constants.
// The generic loaders (LoaderKb / LoaderMNt::load) recompute 5-10 vector instructions of offsets per request -- ~190 per 128 x 128 tile,
// more than the operand split itself.  Same thread maps / register order as LoaderKb / LoaderMNt (x3_store_* and rowsum_finish rely on it).
template <int BMN>
struct X3LoaderK {              // [mn][k], k contiguous: thread t -> rows (t >> 3) + 32 i, k offset (t & 7) * 4
    static constexpr int NV = BMN / 32;
    const float *base;
    long long bytes;            // extent of the operand from `base`
    unsigned voff[NV];          // row offset + 4 * kq, BUF_OOB for rows outside the operand
    int kq;
    bool vec;
    __device__ __forceinline__ void init(const float *p, long long ld, int mn0, int MN, int K, bool vec_, int tid) {
        base = p; bytes = ((long long)(MN - 1) * ld + K) * 4; vec = vec_;
        kq = (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int g = mn0 + (tid >> 3) + 32 * i;
            voff[i] = g < MN ? (unsigned)((long long)g * ld * 4) + 4u * (unsigned)kq : BUF_OOB;
        }
    }
    // tile [k0, k0 + 32) of the range that ends at kend (k0 wave-uniform).  vec: K % 4 == 0, a float4 is inside or outside as a whole
    __device__ __forceinline__ void load(int k0, int kend, float4 (&r)[NV]) const {
        long long left = k0 < kend ? bytes - 4ll * k0 : 0;
        left = left < 0 ? 0 : left;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + k0), 0, (int)(unsigned)left, 0x00020000);
        const int nvalid = kend - (k0 + kq);
        if (vec) {
            const bool inside = nvalid >= 4;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, inside ? voff[i] : BUF_OOB, 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool ok = voff[i] != BUF_OOB;
                r[i].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && nvalid > 0) ? voff[i] : BUF_OOB, 0, 0));
                r[i].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && nvalid > 1) ? voff[i] + 4u : BUF_OOB, 0, 0));
                r[i].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && nvalid > 2) ? voff[i] + 8u : BUF_OOB, 0, 0));
                r[i].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && nvalid > 3) ? voff[i] + 12u : BUF_OOB, 0, 0));
            }
        }
    }
};
template <int BMN>
struct X3LoaderMN {             // [k][mn], mn contiguous: LoaderMNt's unit map (u = t + 256 i: c = u & 3, kr = (u >> 2) & 3, ib, kb)
    static constexpr int NB = BMN / 16;
    static constexpr int NV = BMN / 32;
    const float *base;
    long long ld;
    int MN;
    unsigned voff[NV];          // kr * ld * 4 + col * 4 (vec: BUF_OOB when the float4 leaves the row)
    int ncol[NV];               // scalar path: columns left from `col`
    bool vec;
    __device__ __forceinline__ void init(const float *p, long long ld_, int mn0, int MN_, int K, bool vec_, int tid) {
        base = p; ld = ld_; MN = MN_; vec = vec_;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = tid + 256 * i;
            const int kr = 4 * (u / (16 * NB)) + ((u >> 2) & 3);
            const int col = mn0 + 16 * ((u >> 4) & (NB - 1)) + 4 * (u & 3);
            ncol[i] = MN - col;
            voff[i] = (ncol[i] >= (vec ? 4 : 1)) ? (unsigned)((long long)kr * ld * 4) + 4u * (unsigned)col : BUF_OOB;
        }
    }
    // the descriptor ends with row kend - 1: rows at or past kend fall outside by themselves (ld >= MN)
    __device__ __forceinline__ void load(int k0, int kend, float4 (&r)[NV]) const {
        long long left = ((long long)(kend - 1 - k0) * ld + MN) * 4;
        left = (k0 < kend && left > 0) ? left : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + (long long)k0 * ld), 0, (int)(unsigned)left, 0x00020000);
        if (vec) {
#pragma unroll
            for (int i = 0; i < NV; ++i) r[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const bool ok = voff[i] != BUF_OOB;
                r[i].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? voff[i] : BUF_OOB, 0, 0));
                r[i].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && ncol[i] > 1) ? voff[i] + 4u : BUF_OOB, 0, 0));
                r[i].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && ncol[i] > 2) ? voff[i] + 8u : BUF_OOB, 0, 0));
                r[i].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && ncol[i] > 3) ? voff[i] + 12u : BUF_OOB, 0, 0));
            }
        }
    }
};

// one float4 of a loader register -> its three bf16 images (4 values = 2 pairs = one 8-byte LDS store per image)
__device__ __forceinline__ void x3_split4(const float4 &v, uint2 &h, uint2 &m, uint2 &l) {
    f32x2_t p0, p1;
    p0[0] = v.x; p0[1] = v.y; p1[0] = v.z; p1[1] = v.w;
    split3_pair(p0, h.x, m.x, l.x);
    split3_pair(p1, h.y, m.y, l.y);
}
// K-contiguous operand (LoaderKb's thread map: rows (t >> 3) + 32 i, k offset (t & 7) * 4)
template <int BMN>
__device__ __forceinline__ void x3_store_k(unsigned short (*S)[BMN][X3_LD], const float4 (&r)[BMN / 32], int tid) {
    const int kq = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < BMN / 32; ++i) {
        const int row = (tid >> 3) + 32 * i;
        uint2 h, m, l;
        x3_split4(r[i], h, m, l);
        *reinterpret_cast<uint2 *>(&S[0][row][kq]) = h;
        *reinterpret_cast<uint2 *>(&S[1][row][kq]) = m;
        *reinterpret_cast<uint2 *>(&S[2][row][kq]) = l;
    }
}
// MN-contiguous operand (LoaderMNt's unit map: the transpose-read image is written linearly, 8 bytes per unit t + 256 i)
template <int BMN>
__device__ __forceinline__ void x3_store_mn(unsigned short (*S)[BMN][X3_LD], const float4 (&r)[BMN / 32], int tid) {
#pragma unroll
    for (int i = 0; i < BMN / 32; ++i) {
        uint2 h, m, l;
        x3_split4(r[i], h, m, l);
        const int o = (tid + 256 * i) * 4;
        *reinterpret_cast<uint2 *>(&S[0][0][0] + o) = h;
        *reinterpret_cast<uint2 *>(&S[1][0][0] + o) = m;
        *reinterpret_cast<uint2 *>(&S[2][0][0] + o) = l;
    }
}

// one register of a loader (unit i of the tile) -> the three images: the pipelined loop interleaves these with its MFMAs
template <int BMN, bool KC>
__device__ __forceinline__ void x3_store_unit(unsigned short (*S)[BMN][X3_LD], const float4 &r, int i, int tid) {
    uint2 h, m, l;
    x3_split4(r, h, m, l);
    unsigned short *p0, *p1, *p2;
    if constexpr (KC) {
        const int row = (tid >> 3) + 32 * i, kq = (tid & 7) * 4;
        p0 = &S[0][row][kq]; p1 = &S[1][row][kq]; p2 = &S[2][row][kq];
    } else {
        const int o = (tid + 256 * i) * 4;
        p0 = &S[0][0][0] + o; p1 = &S[1][0][0] + o; p2 = &S[2][0][0] + o;
    }
    *reinterpret_cast<uint2 *>(p0) = h;
    *reinterpret_cast<uint2 *>(p1) = m;
    *reinterpret_cast<uint2 *>(p2) = l;
}

template <int BMN, bool KC>
__device__ __forceinline__ Split3Frag x3_frag(const unsigned short (*S)[BMN][X3_LD], int row_base, int ks, int lane) {
    Split3Frag f;
    if constexpr (KC) {
        const int r = row_base + (lane & 31), k = ks + (lane >> 5) * 8;
        f.h = *reinterpret_cast<const bf16x8_t *>(&S[0][r][k]);
        f.m = *reinterpret_cast<const bf16x8_t *>(&S[1][r][k]);
        f.l = *reinterpret_cast<const bf16x8_t *>(&S[2][r][k]);
    } else {
        f.h = __builtin_bit_cast(bf16x8_t, frag_tr<BMN>(S[0], row_base, ks, lane));
        f.m = __builtin_bit_cast(bf16x8_t, frag_tr<BMN>(S[1], row_base, ks, lane));
        f.l = __builtin_bit_cast(bf16x8_t, frag_tr<BMN>(S[2], row_base, ks, lane));
    }
    return f;
}

template <int BM, int BN, bool AK, bool BKC, bool DB>
__device__ __forceinline__ void gemm_x3_body(const GemmArgs &g, const int id, const int zidx) {
    constexpr int WGM = 2, WGN = 2, NBUF = DB ? 2 : 1;
    using T = TileCfg<BM, BN, WGM, WGN>;
    __shared__ __attribute__((aligned(16))) char smem_raw[X3SmemBytes<BM, BN, NBUF>::VALUE];
    X3Smem<BM, BN, NBUF> &sm = *reinterpret_cast<X3Smem<BM, BN, NBUF> *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tn = id % g.tiles_n, tm = id / g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = zidx % g.split_k;
    const int zb = zidx / g.split_k;
    const int z0 = zb / g.batch_inner, z1 = zb % g.batch_inner;
    const float *A = g.A + z0 * g.sA0 + z1 * g.sA1;
    const float *B = g.B + z0 * g.sB0 + z1 * g.sB1;
    float *C = g.C + z0 * g.sC0 + z1 * g.sC1 + (long long)split * g.part_stride;
    // split ranges in units of 32 k (the host's gemm_effective_split counts the same tiles for this kernel)
    const int nkt = (g.K + X3_BK - 1) / X3_BK;
    const int per = (nkt + g.split_k - 1) / g.split_k;
    const int kt0 = split * per;
    const int kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;
    const int kend = min(g.K, kt1 * X3_BK);

    using LA = typename std::conditional<AK, X3LoaderK<BM>, X3LoaderMN<BM>>::type;
    using LB = typename std::conditional<BKC, X3LoaderK<BN>, X3LoaderMN<BN>>::type;
    LA la;
    LB lb;
    la.init(A, g.lda, m0, g.M, g.K, g.a_vec != 0, tid);
    lb.init(B, g.ldb, n0, g.N, g.K, g.b_vec != 0, tid);

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const bool do_rs = !AK && g.rowsum != nullptr && tn == 0;      // workgroup-uniform
    float4 rs[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
    constexpr int NA = BM / 32, NB_ = BN / 32;
    auto rs_add = [&](const float4 (&r)[NA]) {
#pragma unroll
        for (int i = 0; i < NA; ++i) { rs[0].x += r[i].x; rs[0].y += r[i].y; rs[0].z += r[i].z; rs[0].w += r[i].w; }
    };
    if constexpr (!DB) {
        // one LDS buffer, one register set: [split + store tile t] [request tile t+1] barrier [MFMA phase] barrier; the co-resident
        // workgroups of the CU cover each other's split phase
        float4 ra[NA], rb[NB_];
        la.load(kt0 * X3_BK, kend, ra);
        lb.load(kt0 * X3_BK, kend, rb);
        for (int kt = kt0; kt < kt1; ++kt) {
            if (do_rs) rs_add(ra);
            if constexpr (AK) x3_store_k<BM>(sm.A[0], ra, tid);
            else x3_store_mn<BM>(sm.A[0], ra, tid);
            if constexpr (BKC) x3_store_k<BN>(sm.B[0], rb, tid);
            else x3_store_mn<BN>(sm.B[0], rb, tid);
            // the next tile's requests fly under this tile's MFMA phase (unconditional: past the range they resolve to the out-of-range offset)
            if constexpr ((DETR_X3_ABLATE & 4) == 0) {
                la.load((kt + 1) * X3_BK, kend, ra);
                lb.load((kt + 1) * X3_BK, kend, rb);
            }
            lds_barrier();
#pragma unroll
            for (int ks = 0; ks < X3_BK; ks += 16) {
                Split3Frag a[T::TM], b[T::TN];
#pragma unroll
                for (int mi = 0; mi < T::TM; ++mi) a[mi] = x3_frag<BM, AK>(sm.A[0], wm * T::WTM + mi * 32, ks, lane);
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni) b[ni] = x3_frag<BN, BKC>(sm.B[0], wn * T::WTN + ni * 32, ks, lane);
#pragma unroll
                for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < T::TN; ++ni) acc[mi][ni] = split3_mma<DETR_SPLIT3_TERMS>(a[mi], b[ni], acc[mi][ni]);
            }
            lds_barrier();
        }
    } else {
        // two LDS buffers, two register sets, ONE barrier per tile: while tile t is multiplied out of LDS[cur], tile t+1 (in one register
        // set since the previous iteration) is split and stored into LDS[cur ^ 1] BETWEEN the MFMAs of the same wave -- one unit (float4:
        // 18 VALU + 3 LDS stores) per group of MFMAs -- and tile t+2 is requested into the other register set.  One workgroup per CU
        // (120 KB at 128 x 128): the overlap is inside the wave, not between workgroups.
        float4 ra0[NA], rb0[NB_], ra1[NA], rb1[NB_];
        la.load(kt0 * X3_BK, kend, ra0);
        lb.load(kt0 * X3_BK, kend, rb0);
        la.load((kt0 + 1) * X3_BK, kend, ra1);
        lb.load((kt0 + 1) * X3_BK, kend, rb1);
        if (do_rs) rs_add(ra0);
#pragma unroll
        for (int i = 0; i < NA; ++i) x3_store_unit<BM, AK>(sm.A[0], ra0[i], i, tid);
#pragma unroll
        for (int i = 0; i < NB_; ++i) x3_store_unit<BN, BKC>(sm.B[0], rb0[i], i, tid);
        lds_barrier();
        constexpr int NUNIT = NA + NB_;                    // units of the next tile to place between this tile's MFMA groups
        constexpr int NGRP = 2 * T::TM * T::TN;            // MFMA groups of one tile: (k-step, mi, ni), 6 (9) MFMAs each
        constexpr int UPG = (NUNIT + NGRP - 1) / NGRP;     // units per MFMA group (1 at 128 x 128, 2 at 64 x 64)
        // iteration: LDS[cur] holds tile kt, `rpa / rpb` hold tile kt+1 (stored now), `rqa / rqb` receive tile kt+2
        auto iter = [&](const int kt, const int cur, float4 (&rpa)[NA], float4 (&rpb)[NB_], float4 (&rqa)[NA], float4 (&rqb)[NB_]) {
            la.load((kt + 2) * X3_BK, kend, rqa);
            lb.load((kt + 2) * X3_BK, kend, rqb);
            if (do_rs && kt + 1 < kt1) rs_add(rpa);
            int grp = 0;
#pragma unroll
            for (int ks = 0; ks < X3_BK; ks += 16) {
                Split3Frag a[T::TM], b[T::TN];
#pragma unroll
                for (int mi = 0; mi < T::TM; ++mi) a[mi] = x3_frag<BM, AK>(sm.A[cur], wm * T::WTM + mi * 32, ks, lane);
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni) b[ni] = x3_frag<BN, BKC>(sm.B[cur], wn * T::WTN + ni * 32, ks, lane);
#pragma unroll
                for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < T::TN; ++ni) {
                        acc[mi][ni] = split3_mma<DETR_SPLIT3_TERMS>(a[mi], b[ni], acc[mi][ni]);
#pragma unroll
                        for (int q = 0; q < UPG; ++q) {
                            const int u = grp * UPG + q;
                            if (u < NA) x3_store_unit<BM, AK>(sm.A[cur ^ 1], rpa[u < NA ? u : 0], u, tid);
                            else if (u < NUNIT) x3_store_unit<BN, BKC>(sm.B[cur ^ 1], rpb[u < NUNIT ? u - NA : 0], u - NA, tid);
                        }
                        ++grp;
                    }
            }
            lds_barrier();
        };
        int kt = kt0;
        for (; kt + 2 <= kt1; kt += 2) {
            iter(kt, 0, ra1, rb1, ra0, rb0);
            iter(kt + 1, 1, ra0, rb0, ra1, rb1);
        }
        if (kt < kt1) iter(kt, 0, ra1, rb1, ra0, rb0);
    }
    __syncthreads();
    if (do_rs) rowsum_finish<BM, 1>(rs, reinterpret_cast<float *>(smem_raw), g, m0, split, tid, 1);
    if (g.slab_ts) {
        store_slab_ts<BM, BN, WGM, WGN>(acc, C + (long long)id * (BM * BN), wave, lane);
        return;
    }
    epilogue<BM, BN, WGM, WGN, false>(acc, reinterpret_cast<float *>(smem_raw), C, g.ldc, g.M, g.N, m0, n0, wm, wn, lane, wave, g.e);
}

template <int BM, int BN, bool AK, bool BKC, bool DB>
__global__ __launch_bounds__(GEMM_THREADS, (BM * BN >= 128 * 128) ? (DB ? 1 : 2) : (BM * BN >= 128 * 64 ? (DB ? 1 : 3) : (DB ? 2 : 4))) void gemm_x3_kernel(GemmArgs g) {
    int tile, z;
    gemm_work_item(g, tile, z);
    gemm_x3_body<BM, BN, AK, BKC, DB>(g, tile, z);
}
template <int BM, int BN, bool AK, bool BKC>
__global__ __launch_bounds__(GEMM_THREADS, 4) void gemm_x3_group_kernel(GemmGroupArgs G) {
    int m, tile, z;
    if (!gemm_group_item(G, m, tile, z)) return;
    gemm_x3_body<BM, BN, AK, BKC, false>(G.g[m], tile, z);
}

}  // namespace detr
