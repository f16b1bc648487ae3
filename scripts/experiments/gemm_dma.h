// NOT part of the product build (round 3 experiment, measured and not kept: profiles/r03_micro_gemm_dma_experiment.txt).
// It was wired into csrc/gemm_f32.hip by commit 8b97606 (gemm_bf16dma_kernel) and taken out again.
// gemm_dma.h -- operand path of the bf16 tile engine for operands that are ALREADY bf16 in memory: global -> LDS by
// LDS-DMA (`buffer_load_dwordx4 ... lds`), no VGPR staging and no ds_write pass.
//
// Why (round 3): per 32-deep K tile of a 64x64 tile the register-staged loop moves 8 KB through
// ds_write_b128 (~79 B/clk/CU: ~104 LDS-pipe cycles) and 16 KB through ds_read_b128 (64 cycles) against 64 cycles of MFMA per
// SIMD -- the LDS store path alone is longer than the MFMA work (MI355X_MICROARCH.md, LDS table), and the two register sets of
// the two-deep pipeline are the kernel's VGPR budget.  An LDS-DMA piece writes 1 KB per wave instruction straight into LDS at
// the array's width; a ring of NS stages replaces the register sets, and a request stays in flight across the (raw)
// barriers until a counted `s_waitcnt vmcnt(N)` in front of the barrier that publishes its stage.
//
// The DMA writes lane-linearly: lane s of a piece lands at byte 16*s of the piece.  The LDS image is therefore chosen by
// which GLOBAL address lane s requests (cdna_hip_programming.md rule 21: swizzle the source, never the destination):
//   * K-contiguous operand ([mn][k]): rows of BK bf16 = CPR = BK/8 sixteen-byte chunks, unpadded; chunk c of row r sits at
//     slot CPR*r + (c ^ f(r)), f(r) = (r >> 2) & 3 for BK = 32 (64-byte rows), (r >> 1) & 7 for BK = 64 (128-byte rows): the 16
//     lanes of every ds_read_b128 service group (rows distinct mod 16, one chunk index) hit 16 distinct 16-byte slots of
//     the 256-byte bank row -- conflict free, no padding (which a lane-linear DMA could not produce).
//   * MN-contiguous operand ([k][mn]): the transpose-read image of gemm_bf16_core.h unchanged ([4 k][16 mn] sub-blocks of
//     128 B, sub-block (kb, ib) at (kb * NB + ib) * 128 B, fragments by ds_read_b64_tr_b16): a 16-byte slot is 8 consecutive
//     mn of one k row, which is exactly one DMA lane; a piece is 8 consecutive sub-blocks.
// Out-of-range lanes (row >= MN, k >= this split's K end) request the buffer descriptor's out-of-range offset: the DMA
// writes zeros (the hardware's bounds check; composable_kernel's direct loads rely on the same behaviour).
//
// Pipeline (per workgroup, 4 waves, every wave issues its share of the pieces of both operands):
//     prologue   issue tiles 0 .. NS-2
//     iteration  s_waitcnt vmcnt((NS-2) * pieces per tile and wave)   -> this wave's pieces of tile t have landed
//                s_barrier (raw: no vmcnt drain)                     -> everybody's have; everybody is done reading tile t-1
//                issue tile t+NS-1 into the stage tile t-1 occupied
//                fragments of tile t out of LDS, MFMAs
// Requests are unconditional (past the last tile they are out of range and fill a stage nobody reads), so the counts
// in the waits are exact on every path.
#pragma once
#include "gemm_bf16_core.h"

namespace detr {

typedef __attribute__((address_space(3))) void lds_void_t;

template <int BK>
struct DmaK {      // K-contiguous image: chunk swizzle
    static constexpr int CPR = BK / 8;                       // 16-byte chunks per row
    static constexpr int ROWS_PER_PIECE = 64 / CPR;          // rows one 1 KB piece covers
    __device__ __forceinline__ static int f(int r) { return BK == 32 ? ((r >> 2) & 3) : ((r >> 1) & 7); }
    __device__ __forceinline__ static int slot(int r, int c) { return CPR * r + (c ^ f(r)); }
};

// One operand's requests.  KC: K-contiguous ([mn][k]); otherwise MN-contiguous ([k][mn]).  NP = pieces per wave and K tile.
template <int BMN, int BK, bool KC>
struct DmaLoader {
    static constexpr int PIECES = BMN * BK * 2 / 1024;       // 1 KB pieces per tile
    static constexpr int NP = PIECES / 4;                    // per wave
    static_assert(PIECES % 4 == 0 && NP >= 1, "tile too small for four waves of 1 KB pieces");
    static constexpr int NB = BMN / 16;                      // (MN-contiguous) sub-blocks per k group
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned base[NP];        // KC: row byte offset + 16 * source chunk (BUF_OOB: row outside);  MN: column byte offset (BUF_OOB: outside)
    int kk[NP];               // KC: k of the chunk's END relative to the tile (8c + 8);         MN: k of this lane's row relative to the tile
    unsigned ld2b;            // MN: row stride in bytes
    int piece0;               // first piece of this wave

    __device__ __forceinline__ void init(const void *p, long long ld, int mn0, int MN, int K, int lane, int wave) {
        const long long extent = KC ? ((long long)(MN - 1) * ld + K) : ((long long)(K - 1) * ld + MN);
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(unsigned)(extent * 2), 0x00020000);
        ld2b = (unsigned)(ld * 2);
        piece0 = wave * NP;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int P = piece0 + i;
            if constexpr (KC) {
                using D = DmaK<BK>;
                const int r = P * D::ROWS_PER_PIECE + lane / D::CPR;       // tile row this lane fills
                const int c = (lane % D::CPR) ^ D::f(r);                   // source chunk that belongs at this lane's slot
                const int g = mn0 + r;
                base[i] = g < MN ? (unsigned)((long long)g * ld * 2) + 16u * (unsigned)c : BUF_OOB;
                kk[i] = 8 * c + 8;
            } else {
                const int sb = P * 8 + (lane >> 3);                         // sub-block: kb = sb / NB, ib = sb % NB
                const int kb = sb / NB, ib = sb % NB;
                const int col = mn0 + 16 * ib + 8 * (lane & 1);
                base[i] = (col + 8 <= MN) ? 2u * (unsigned)col : BUF_OOB;
                kk[i] = 4 * kb + ((lane >> 1) & 3);
            }
        }
    }
    // request tile [k0, k0 + BK) of this split (k < kend) into `stage` (wave-uniform LDS address of the operand's stage)
    __device__ __forceinline__ void issue(int k0, int kend, unsigned short *stage) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            unsigned voff;
            if constexpr (KC) voff = (base[i] != BUF_OOB && k0 + kk[i] <= kend) ? base[i] + 2u * (unsigned)k0 : BUF_OOB;
            else voff = (base[i] != BUF_OOB && k0 + kk[i] < kend) ? (unsigned)(k0 + kk[i]) * ld2b + base[i] : BUF_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t *)(stage + (piece0 + i) * 512), 16, voff, 0, 0, 0);
        }
    }
};

// fragments of one K tile out of the DMA images + MFMAs (the counterpart of mma_ktile_bf16)
template <int BM, int BN, int WGM, int WGN, int BK, bool AK, bool BKC>
__device__ __forceinline__ void mma_ktile_dma(const unsigned short *As, const unsigned short *Bs,
                                              f32x16 (&acc)[TileCfg<BM, BN, WGM, WGN>::TM][TileCfg<BM, BN, WGM, WGN>::TN],
                                              int wm, int wn, int lane) {
    using T = TileCfg<BM, BN, WGM, WGN>;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
        bf16x8 a[T::TM], b[T::TN];
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi) {
            const int row = wm * T::WTM + mi * 32;
            if constexpr (AK) a[mi] = *reinterpret_cast<const bf16x8 *>(As + 8 * DmaK<BK>::slot(row + l31, ks / 8 + h));
            else a[mi] = frag_tr<BM>(reinterpret_cast<const unsigned short (*)[BF_LD]>(As), row, ks, lane);
        }
#pragma unroll
        for (int ni = 0; ni < T::TN; ++ni) {
            const int row = wn * T::WTN + ni * 32;
            if constexpr (BKC) b[ni] = *reinterpret_cast<const bf16x8 *>(Bs + 8 * DmaK<BK>::slot(row + l31, ks / 8 + h));
            else b[ni] = frag_tr<BN>(reinterpret_cast<const unsigned short (*)[BF_LD]>(Bs), row, ks, lane);
        }
#pragma unroll
        for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
    }
}

template <int BM, int BN, int BK, int NS, int WGN>
struct DmaSmemBytes {
    static constexpr int STAGE_A = BM * BK * 2, STAGE_B = BN * BK * 2;
    static constexpr int TILES = NS * (STAGE_A + STAGE_B);
    static constexpr int EPI = 4 * 32 * (BN / WGN + 4) * 4;
    static constexpr int VALUE = TILES > EPI ? TILES : EPI;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {          // counted wait on the VM counter only (LDS-DMA requests count there)
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// K loop of one (tile, split): accumulates into acc.  smem: the workgroup's ONE shared array (the epilogue reuses it).
template <int BM, int BN, int WGM, int WGN, int BK, int NS, bool AK, bool BKC>
__device__ __forceinline__ void gemm_dma_kloop(const void *A, long long lda, const void *B, long long ldb, int M, int N, int m0, int n0,
                                               int kbeg, int kend, char *smem,
                                               f32x16 (&acc)[TileCfg<BM, BN, WGM, WGN>::TM][TileCfg<BM, BN, WGM, WGN>::TN]) {
    using SB = DmaSmemBytes<BM, BN, BK, NS, WGN>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    DmaLoader<BM, BK, AK> la;
    DmaLoader<BN, BK, BKC> lb;
    // K extent of the descriptors = this split's end: a request past it is out of range whatever its row
    la.init(A, lda, m0, M, kend, lane, wave);
    lb.init(B, ldb, n0, N, kend, lane, wave);
    constexpr int PER_TILE = DmaLoader<BM, BK, AK>::NP + DmaLoader<BN, BK, BKC>::NP;       // requests per wave and tile
    auto stage_a = [&](int st) { return reinterpret_cast<unsigned short *>(smem + st * (SB::STAGE_A + SB::STAGE_B)); };
    auto stage_b = [&](int st) { return reinterpret_cast<unsigned short *>(smem + st * (SB::STAGE_A + SB::STAGE_B) + SB::STAGE_A); };
    const int nkt = (kend - kbeg + BK - 1) / BK;
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
        la.issue(kbeg + t * BK, kend, stage_a(t));
        lb.issue(kbeg + t * BK, kend, stage_b(t));
    }
    // one tile: compile-time stage indices (the fragment addresses become immediates); the loop runs whole rounds of NS
    // tiles and a tail of < NS -- every executed step waits, publishes, requests and multiplies in the same order, so the
    // counted waits hold on every path
    auto step = [&](auto S, int t) {
        constexpr int s = decltype(S)::value;
        constexpr int nst = (s + NS - 1) % NS;
        wait_vmcnt<(NS - 2) * PER_TILE>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");        // (compiler-level only: neither the requests nor the fragment reads may move above the barrier)
        la.issue(kbeg + (t + NS - 1) * BK, kend, stage_a(nst));
        lb.issue(kbeg + (t + NS - 1) * BK, kend, stage_b(nst));
        mma_ktile_dma<BM, BN, WGM, WGN, BK, AK, BKC>(stage_a(s), stage_b(s), acc, wm, wn, lane);
    };
    int t = 0;
    for (; t + NS <= nkt; t += NS) {
        step(std::integral_constant<int, 0>{}, t);
        if constexpr (NS > 1) step(std::integral_constant<int, 1 % NS>{}, t + 1);
        if constexpr (NS > 2) step(std::integral_constant<int, 2 % NS>{}, t + 2);
        if constexpr (NS > 3) step(std::integral_constant<int, 3 % NS>{}, t + 3);
    }
    static_assert(NS >= 2 && NS <= 4, "ring of 2 .. 4 stages");
    if (t < nkt) step(std::integral_constant<int, 0>{}, t);
    if constexpr (NS > 2) { if (t + 1 < nkt) step(std::integral_constant<int, 1>{}, t + 1); }
    if constexpr (NS > 3) { if (t + 2 < nkt) step(std::integral_constant<int, 2 % NS>{}, t + 2); }
    wait_vmcnt<0>();                              // the trailing out-of-range requests still write (zeros) into the ring:
    __syncthreads();                              // nobody reuses the array before they have landed
}

}  // namespace detr
