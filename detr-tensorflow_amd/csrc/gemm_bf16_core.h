// gemm_bf16_core.h -- bf16-COMPUTE tile engine (fp32 storage, bf16 MFMA, fp32 accumulate) for gfx950.
//
// Used when the engine runs in precision="bf16" (BASELINE.json config C3): operands stay fp32 in HBM and are
// rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way into LDS; v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA
// rate) accumulates in fp32; the epilogue (scale/bias/residual/activation/mask/dropout, float4 stores through
// LDS) is the fp32 one from gemm_core.h.  With fp32 storage these kernels are HBM / L2 bound, not MFMA bound.
//
// LDS image: [row][k] with k contiguous, 32 k per tile, row stride 40 bf16 = 80 B.  A 32x32x16 fragment is 8
// consecutive k of one row = one ds_read_b128; 80 B = 5 sixteen-byte slots per row and gcd(5,16) = 1, so the 16
// lanes of every ds_read_b128 service group (distinct rows mod 16) hit 16 distinct slots: conflict free.
//   * K-contiguous fp32 operand ([mn][k]): float4 global load -> 4 bf16 -> one ds_write_b64.
//   * MN-contiguous fp32 operand ([k][mn]: weight-gradient operands, Linear dgrad weights, HWIO conv kernels, the
//     gathered pixels of the conv weight gradient) (LoaderMNt / LoaderWgradAt): the tile keeps its natural orientation in LDS, cut into [4 k][16 mn] sub-blocks of 128 B, and
//     the MFMA fragments are fetched with ds_read_b64_tr_b16 (gfx950 transpose read: the 16 lanes of a group hand
//     in the sixteen 8-byte chunks of one sub-block and lane c receives column c = 4 consecutive k of one row).
//     Global loads are float4 along mn (256 B contiguous per k row and wave), the LDS write is one linear
//     ds_write_b64 per float4 -- the same cost as the K-contiguous path; fragment = 2 tr reads.
#pragma once
#include "gemm_core.h"

namespace detr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BF_BK = 32;
constexpr int BF_LD = 40;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 r;
    r[0] = (__bf16)a;
    r[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, r);
}

// BK: depth of one K tile (32, or 64 for the all-bf16 long-K variant: half the barrier-separated iterations per K);
// rows are padded by 8 bf16 (16 B): row r starts at 16-byte slot (BK/8 + 1) * r -- 5 r or 9 r, both odd, so 16 consecutive
// rows fall into 16 different slots mod 16 and the b128 fragment reads are conflict-free
template <int BM, int BN, int BK = BF_BK>
struct BfSmem {
    unsigned short A[2][BM][BK + 8];
    unsigned short B[2][BN][BK + 8];
};

template <int BM, int BN, int WGN, int BK = BF_BK>
struct BfSmemBytes {
    static constexpr int TILES = (int)sizeof(BfSmem<BM, BN, BK>);
    static constexpr int STAGE = 4 * 32 * (BN / WGN + 4) * 4;
    static constexpr int VALUE = TILES > STAGE ? TILES : STAGE;
};

// fp32 operand stored [mn][k] (k contiguous). Thread t: rows (t>>3) + 32*i, k offset (t&7)*4.
template <int BMN>
struct LoaderKb {
    static constexpr int NV = BMN / 32;
    typedef float4 Reg;
    static constexpr int NREG = NV;
    static constexpr int NDSW = NV, NVMEM = NV;      // DS write / (vector-path) request instructions per tile
    BufSrc src;
    unsigned off[NV];      // byte offset of the row, BUF_OOB for rows outside the operand
    bool vec;
    int kq, tid;

    __device__ __forceinline__ void init(const float *p, long long ld, int mn0, int MN, int K, bool vec_, int tid_,
                                         long long extent_elems = 0) {
        src.init(p, extent_elems > 0 ? extent_elems : (long long)(MN - 1) * ld + K);
        vec = vec_; tid = tid_;
        kq = (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int g = mn0 + (tid >> 3) + 32 * i;
            off[i] = g < MN ? (unsigned)((long long)g * ld * 4) : BUF_OOB;
        }
    }
    __device__ __forceinline__ void load(int k0, int K, float4 (&r)[NV], unsigned base = 0) const {
        const int k = k0 + kq;
        if (vec) {       // wave-uniform: one scalar branch per tile load, none per float4
#pragma unroll
            for (int i = 0; i < NV; ++i)
                r[i] = src.ld4_vec(off[i] + base + 4u * (unsigned)k, off[i] != BUF_OOB ? K - k : 0);
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                r[i] = src.ld4_scalar(off[i] + base + 4u * (unsigned)k, off[i] != BUF_OOB ? K - k : 0);
        }
    }
    __device__ __forceinline__ void store(unsigned short (*S)[BF_LD], const float4 (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = (tid >> 3) + 32 * i;
            *reinterpret_cast<uint2 *>(&S[row][kq]) = make_uint2(pack_bf16(r[i].x, r[i].y), pack_bf16(r[i].z, r[i].w));
        }
    }
};

// fp32 operand stored [k][mn] (mn contiguous), transpose-read image.  Unit u = t + 256*i, bit fields (low to high):
// c = u & 3 (float4 inside a sub-block row), kr = (u >> 2) & 3 (k inside the sub-block), ib = mn sub-block, kb = k
// sub-block; the LDS image is written LINEARLY in u (8 bytes per unit), sub-block (kb, ib) at ((kb * NB) + ib) * 128 B.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int BMN>
struct LoaderMNt {
    static constexpr int NB = BMN / 16;
    static constexpr int NU = BMN / 32;              // float4 per thread and tile (32 k x BMN / 4 / 256)
    typedef float4 Reg;
    static constexpr int NREG = NU;
    static constexpr int NDSW = NU, NVMEM = NU;
    BufSrc src;
    unsigned ld4b;
    int mn0, MN;
    bool vec;
    int tid;

    __device__ __forceinline__ void init(const float *p, long long ld_, int mn0_, int MN_, int K, bool vec_, int tid_,
                                         long long extent_elems = 0) {
        src.init(p, extent_elems > 0 ? extent_elems : (long long)(K - 1) * ld_ + MN_);
        ld4b = (unsigned)(ld_ * 4); mn0 = mn0_; MN = MN_; vec = vec_; tid = tid_;
    }
    __device__ __forceinline__ void load(int k0, int K, float4 (&r)[NU], unsigned base = 0) const {
        unsigned o[NU];
        int nv[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = tid + 256 * i;
            const int k = k0 + 4 * (u / (16 * NB)) + ((u >> 2) & 3);
            const int col = mn0 + 16 * ((u >> 4) & (NB - 1)) + 4 * (u & 3);
            o[i] = base + (unsigned)k * ld4b + 4u * (unsigned)col;
            nv[i] = k < K ? MN - col : 0;
        }
        if (vec) {
#pragma unroll
            for (int i = 0; i < NU; ++i) r[i] = src.ld4_vec(o[i], nv[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) r[i] = src.ld4_scalar(o[i], nv[i]);
        }
    }
    __device__ __forceinline__ void store(unsigned short (*S)[BF_LD], const float4 (&r)[NU]) const {
        unsigned short *flat = &S[0][0];
#pragma unroll
        for (int i = 0; i < NU; ++i)
            *reinterpret_cast<uint2 *>(flat + (tid + 256 * i) * 4) = make_uint2(pack_bf16(r[i].x, r[i].y), pack_bf16(r[i].z, r[i].w));
    }
};

// ---- operands that are ALREADY bf16 in memory (the per-step bf16 shadow of the weights, engine.py): no conversion,
// half the bytes; same LDS images as LoaderKb / LoaderMNt.  K (k-contiguous) must be a multiple of 8, MN of 4.
// [mn][k], k contiguous: 16-byte chunks of 8 k, CPR = BK / 8 of them per row; thread t: rows t / CPR + (256 / CPR) * i, k offset
// (t % CPR) * 8  (BK = 32: rows (t >> 2) + 64 i)
template <int BMN, int BK = BF_BK>
struct LoaderKh {
    static constexpr int CPR = BK / 8;
    static constexpr int RPP = 256 / CPR;             // rows per pass of the 256 threads
    static constexpr int NV = (BMN + RPP - 1) / RPP;
    static constexpr bool PART = BMN % RPP != 0;      // a tile with fewer rows than one pass (32 rows x 32 k): the upper threads idle
    static_assert(!PART || NV == 1, "tile rows: a multiple of the rows one pass covers, or fewer than one pass");
    typedef uint4 Reg;
    static constexpr int NREG = NV;
    static constexpr int NDSW = NV, NVMEM = NV;
    BufSrc src;
    unsigned off[NV];
    int k8, tid;
    __device__ __forceinline__ void init(const float *p, long long ld, int mn0, int MN, int K, bool, int tid_,
                                         long long extent_elems = 0) {
        src.init_bytes(p, (extent_elems > 0 ? extent_elems : (long long)(MN - 1) * ld + K) * 2);
        tid = tid_;
        k8 = (tid % CPR) * 8;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int lr = tid / CPR + RPP * i;
            const int g = mn0 + lr;
            off[i] = (g < MN && (!PART || lr < BMN)) ? (unsigned)((long long)g * ld * 2) : BUF_OOB;
        }
    }
    __device__ __forceinline__ void load(int k0, int K, uint4 (&r)[NV], unsigned base = 0) const {
        const int k = k0 + k8;
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = src.ld16((off[i] != BUF_OOB && k + 8 <= K) ? off[i] + base + 2u * (unsigned)k : BUF_OOB);
    }
    template <int LD>
    __device__ __forceinline__ void store(unsigned short (*S)[LD], const uint4 (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (!PART || tid / CPR + RPP * i < BMN) *reinterpret_cast<uint4 *>(&S[tid / CPR + RPP * i][k8]) = r[i];
    }
    // ---- per-tile descriptor form (round 4): the K advance lives in the DESCRIPTOR (base += 2 k0, num_records -= 2 k0: scalar
    // instructions), the per-lane offsets voff[] = off[] + 2 k8 are loop constants, and a tile at or past `kend` gets an empty
    // descriptor (no traffic).  What is left per tile is one compare + one select per request for a ragged last tile --
    // the offset arithmetic of load() was ~5 VALU instructions per request, more than the tile's MFMAs leave room for.
    const unsigned short *tbase;
    long long text;
    __device__ __forceinline__ void init_tiles(const float *p, long long ld, int MN, int K) {
        tbase = reinterpret_cast<const unsigned short *>(p);
        text = ((long long)(MN - 1) * ld + K) * 2;
    }
    __device__ __forceinline__ void load_tile(int k0, int kend, uint4 (&r)[NV]) const {
        long long left = (k0 < kend) ? text - 2ll * k0 : 0;
        left = left < 0 ? 0 : left;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(tbase + k0), 0, (int)(unsigned)left, 0x00020000);
        const bool inside = k0 + k8 < kend;            // (K % 8 == 0: a 16-byte chunk is inside or outside as a whole)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned vo = off[i] != BUF_OOB ? off[i] + 2u * (unsigned)k8 : BUF_OOB;      // loop constant
            r[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, inside ? vo : BUF_OOB, 0, 0));
        }
    }
};

// [k][mn], mn contiguous: transpose-read image, 8-byte units of 4 mn (same unit map as LoaderMNt).
// WIDE (compile time -- a run-time switch around the requests would cost the operand pipelines their exact vmcnt waits,
// measured +0.8 ms per step): a thread requests PAIRS of adjacent units, one 16-byte load and one ds_write_b128 instead of
// two 8-byte ones; register 2i / 2i + 1 then hold units 2p / 2p + 1 of pair p = tid + 256 i.  Needs MN % 8 == 0, a row stride
// % 8 == 0 and a tile origin % 8 == 0 (checked by the host dispatch); a caller that sums the registers per column group
// (the fused bias gradient of gemm_bf16c_body) uses the narrow form: a pair spans two column groups.
template <int BMN, bool WIDE = false, int BK = BF_BK>
struct LoaderMNth {
    static constexpr int NB = BMN / 16;
    static constexpr int NU = BMN * BK / 1024;       // 8-byte units per thread and tile (BK k x BMN / 4 / 256)
    static_assert(!WIDE || NU % 2 == 0, "wide units come in pairs");
    typedef uint2 Reg;
    static constexpr int NREG = NU;
    static constexpr int NDSW = WIDE ? NU / 2 : NU, NVMEM = WIDE ? NU / 2 : NU;
    BufSrc src;
    unsigned ld2b;
    int mn0, MN, tid;
    __device__ __forceinline__ void init(const float *p, long long ld_, int mn0_, int MN_, int K, bool, int tid_,
                                         long long extent_elems = 0) {
        src.init_bytes(p, (extent_elems > 0 ? extent_elems : (long long)(K - 1) * ld_ + MN_) * 2);
        ld2b = (unsigned)(ld_ * 2); mn0 = mn0_; MN = MN_; tid = tid_;
    }
    __device__ __forceinline__ void load(int k0, int K, uint2 (&r)[NU], unsigned base = 0) const {
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < NU / 2; ++i) {
                const int u = 2 * (tid + 256 * i);
                const int k = k0 + 4 * (u / (16 * NB)) + ((u >> 2) & 3);
                const int col = mn0 + 16 * ((u >> 4) & (NB - 1)) + 4 * (u & 3);
                const uint4 v = src.ld16((k < K && col + 8 <= MN) ? base + (unsigned)k * ld2b + 2u * (unsigned)col : BUF_OOB);
                r[2 * i] = make_uint2(v.x, v.y);
                r[2 * i + 1] = make_uint2(v.z, v.w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int u = tid + 256 * i;
                const int k = k0 + 4 * (u / (16 * NB)) + ((u >> 2) & 3);
                const int col = mn0 + 16 * ((u >> 4) & (NB - 1)) + 4 * (u & 3);
                r[i] = src.ld8((k < K && col + 4 <= MN) ? base + (unsigned)k * ld2b + 2u * (unsigned)col : BUF_OOB);
            }
        }
    }
    // ---- per-tile descriptor form (see LoaderKh): base += k0 rows, num_records shrinks with it, so rows at or past `kend` fall outside
    // the descriptor by themselves (ld >= MN) and the per-lane offsets are loop constants: ZERO vector instructions per request
    const unsigned short *tbase;
    long long text;
    __device__ __forceinline__ void init_tiles(const float *p, long long ld_, int kend) {
        tbase = reinterpret_cast<const unsigned short *>(p);
        text = ((long long)(kend - 1) * ld_ + MN) * 2;          // the descriptor ends with row kend - 1 (this split's last row)
    }
    __device__ __forceinline__ void load_tile(int k0, int, uint2 (&r)[NU]) const {
        long long left = text - (long long)k0 * ld2b;
        left = left < 0 ? 0 : left;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(tbase) + (long long)k0 * (ld2b >> 1), 0,
                                                                           (int)(unsigned)left, 0x00020000);
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < NU / 2; ++i) {
                const int u = 2 * (tid + 256 * i);
                const int kr = 4 * (u / (16 * NB)) + ((u >> 2) & 3);
                const int col = mn0 + 16 * ((u >> 4) & (NB - 1)) + 4 * (u & 3);
                const unsigned vo = (col + 8 <= MN) ? (unsigned)kr * ld2b + 2u * (unsigned)col : BUF_OOB;       // loop constant
                const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
                r[2 * i] = make_uint2(v.x, v.y);
                r[2 * i + 1] = make_uint2(v.z, v.w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int u = tid + 256 * i;
                const int kr = 4 * (u / (16 * NB)) + ((u >> 2) & 3);
                const int col = mn0 + 16 * ((u >> 4) & (NB - 1)) + 4 * (u & 3);
                const unsigned vo = (col + 4 <= MN) ? (unsigned)kr * ld2b + 2u * (unsigned)col : BUF_OOB;
                r[i] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, 0, 0));
            }
        }
    }
    template <int LD>
    __device__ __forceinline__ void store(unsigned short (*S)[LD], const uint2 (&r)[NU]) const {
        unsigned short *flat = &S[0][0];
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < NU / 2; ++i)
                *reinterpret_cast<uint4 *>(flat + (tid + 256 * i) * 8) = make_uint4(r[2 * i].x, r[2 * i].y, r[2 * i + 1].x, r[2 * i + 1].y);
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) *reinterpret_cast<uint2 *>(flat + (tid + 256 * i) * 4) = r[i];
        }
    }
};

// MFMA fragment (8 consecutive k of row `row_base + (lane & 31)`) out of a transpose-read image
template <int BMN, int LD>
__device__ __forceinline__ bf16x8 frag_tr(const unsigned short (*S)[LD], int row_base, int ks, int lane) {
    constexpr int NB = BMN / 16;
    const int g = lane >> 4, t = lane & 15;
    const int ib = (row_base >> 4) + (g & 1);
    const int kb = (ks >> 2) + 2 * (g >> 1);
    const unsigned short *p = &S[0][0] + ((kb * NB + ib) * 64 + t * 4);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + NB * 64));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// ---- explicitly pipelined K tile (round 4; DETR_KLOOP_PIPE) -----------------------------------------------------------------
// SQ counters of the split-K weight-gradient launches (one 4-wave workgroup per CU = ONE wave per SIMD, profiles/r03_rocprofv3_pmc_sq_bf16.txt):
// a K tile costs a wave ~2500 cycles for 512 cycles of MFMA -- 37 % issuing, 33 % parked on s_waitcnt, 30 % issue stalls -- because
// the compiler's order is [LDS stores of the next tile] -> [fragment reads] -> wait -> [MFMAs], i.e. every phase's latency is exposed
// when no second wave is there to cover it.  Here the iteration is written, and pinned with sched_group_barrier, as
//   reads(step 0), reads(step 1) | MFMA(step 0) with the next tile's LDS stores between them | reads(step 2) |
//   MFMA(step 1) with the tile-after-next's global requests between them | reads(step 3) | MFMA(step 2) | MFMA(step 3)
// so that fragment reads run one k-step ahead of the MFMAs that consume them and stores / requests ride in the MFMA shadows.
// Every accumulator sees its k-steps in the same order as before: results are bit-identical to mma_ktile_bf16.
template <int BM, int BN, int WGM, int WGN, bool ATR, bool BTR, int BK>
struct KPipe {
    using T = TileCfg<BM, BN, WGM, WGN>;
    static constexpr int NS = BK / 16;
    static constexpr int READS = T::TM * (ATR ? 2 : 1) + T::TN * (BTR ? 2 : 1);     // DS read instructions per k-step
    static constexpr int MF = T::TM * T::TN;                                           // MFMAs per k-step
    struct Frags { bf16x8 a[T::TM], b[T::TN]; };
    __device__ __forceinline__ static void read(const unsigned short (*As)[BK + 8], const unsigned short (*Bs)[BK + 8], int ks, int wm, int wn,
                                                int lane, Frags &f) {
        const int l31 = lane & 31, kh = (lane >> 5) * 8;
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi) {
            if constexpr (ATR) f.a[mi] = frag_tr<BM>(As, wm * T::WTM + mi * 32, ks, lane);
            else f.a[mi] = *reinterpret_cast<const bf16x8 *>(&As[wm * T::WTM + mi * 32 + l31][ks + kh]);
        }
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni) {
            if constexpr (BTR) f.b[ni] = frag_tr<BN>(Bs, wn * T::WTN + ni * 32, ks, lane);
            else f.b[ni] = *reinterpret_cast<const bf16x8 *>(&Bs[wn * T::WTN + ni * 32 + l31][ks + kh]);
        }
    }
    __device__ __forceinline__ static void mma(const Frags &f, f32x16 (&acc)[T::TM][T::TN]) {
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[mi], f.b[ni], acc[mi][ni], 0, 0, 0);
    }
};
// sched_group_barrier masks (LLVM AMDGPU): 0x2 VALU, 0x8 MFMA, 0x20 VMEM read, 0x100 DS read, 0x200 DS write
template <int N> __device__ __forceinline__ void sgb_ds_read() { if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(0x100, N, 0); }
template <int N> __device__ __forceinline__ void sgb_ds_write() { if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(0x200, N, 0); }
template <int N> __device__ __forceinline__ void sgb_vmem() { if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(0x20, N, 0); }
template <int N> __device__ __forceinline__ void sgb_valu() { if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(0x2, N, 0); }
__device__ __forceinline__ void sgb_mfma() { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); }
// MFMA i of MF gets items [i * TOTAL / MF, (i + 1) * TOTAL / MF) of a TOTAL-item side stream
template <int TOTAL, int MF, int I> struct SgbShare { static constexpr int N = ((I + 1) * TOTAL) / MF - (I * TOTAL) / MF; };
template <int MF, int NW, int NG, int NV, int I = 0>
__device__ __forceinline__ void sgb_mfma_block() {       // MF MFMAs, each followed by its share of NW DS writes, NV VALU and NG VMEM requests
    if constexpr (I < MF) {
        sgb_mfma();
        sgb_ds_write<SgbShare<NW, MF, I>::N>();
        sgb_valu<SgbShare<NV, MF, I>::N>();
        sgb_vmem<SgbShare<NG, MF, I>::N>();
        sgb_mfma_block<MF, NW, NG, NV, I + 1>();
    }
}

// steps 2 .. NS-1 of a pipelined K tile: [reads of the next step] [MF MFMAs], the last step without reads
template <int LEFT, int READS, int MF>
__device__ __forceinline__ void sgb_tail() {
    if constexpr (LEFT > 0) {
        if constexpr (LEFT > 1) sgb_ds_read<READS>();
        sgb_mfma_block<MF, 0, 0, 0>();
        sgb_tail<LEFT - 1, READS, MF>();
    }
}

// one BK-deep K tile: BK / 16 k-steps of v_mfma_f32_32x32x16_bf16 per 32x32 output tile.
// operand map: lane l supplies A[i = l&31][k = 8*(l>>5) .. +7] and B[k = 8*(l>>5) .. +7][j = l&31].
template <int BM, int BN, int WGM, int WGN, bool ATR = false, bool BTR = false, int BK = BF_BK>
__device__ __forceinline__ void mma_ktile_bf16(const unsigned short (*As)[BK + 8], const unsigned short (*Bs)[BK + 8],
                                               f32x16 (&acc)[TileCfg<BM, BN, WGM, WGN>::TM][TileCfg<BM, BN, WGM, WGN>::TN],
                                               int wm, int wn, int lane) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    const int l31 = lane & 31;
    const int kh = (lane >> 5) * 8;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
        bf16x8 a[T::TM], b[T::TN];
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi) {
            if constexpr ((DETR_ABLATE & 16) != 0) a[mi] = __builtin_bit_cast(bf16x8, make_uint4(lane, mi, ks, 0x3f803f80u));
            else if (ATR) a[mi] = frag_tr<BM>(As, wm * T::WTM + mi * 32, ks, lane);
            else a[mi] = *reinterpret_cast<const bf16x8 *>(&As[wm * T::WTM + mi * 32 + l31][ks + kh]);
        }
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni) {
            if constexpr ((DETR_ABLATE & 16) != 0) b[ni] = __builtin_bit_cast(bf16x8, make_uint4(lane, ni, ks, 0x3f803f80u));
            else if (BTR) b[ni] = frag_tr<BN>(Bs, wn * T::WTN + ni * 32, ks, lane);
            else b[ni] = *reinterpret_cast<const bf16x8 *>(&Bs[wn * T::WTN + ni * 32 + l31][ks + kh]);
        }
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) {
                if constexpr ((DETR_ABLATE & 1) != 0) { ablate_keep(a[mi]); ablate_keep(b[ni]); }
                else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
    }
}

}  // namespace detr
