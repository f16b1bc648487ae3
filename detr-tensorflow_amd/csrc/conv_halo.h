// conv_halo.h -- stride-1 3x3 convolution forward / input gradient on bf16 tensors with the input HALO STAGED ONCE
// (included by conv_f32.hip; reference: detr_tf/networks/resnet_backbone.py:98-137, the conv2 of every BottleNeck).
//
// Why (ablation builds, scripts/experiments/ablate.sh, profiles/r02_ablation_tile_kernels.txt): in the implicit-GEMM
// tile kernel the operand REQUESTS and their LDS stores cost 35-45 % of the kernel (C = 64: 108 us with them, 60 us
// without) -- every one of the 9 taps re-fetches its [pixels][32 ch] operand through the vector memory path although
// the taps of a tile read the same (tile + 1 pixel border) input patch, and a 64 / 128 pixel tile re-fetches the whole
// 9 x Ci x Co kernel per 64 / 128 output pixels.  Here a workgroup owns 8 x 32 = 256 output pixels x BN output
// channels: per 32-channel chunk the 10 x 34 pixel patch is staged in LDS once ([pixel][32 ch], 80-byte rows) and the
// nine taps read their MFMA A fragments from it at a constant LDS offset ((dh * 34 + dw) * 80 bytes -- the fragment
// rows of a 32-pixel tile row stay consecutive patch rows, so every ds_read_b128 is conflict free for any tap); only
// the weights stream through LDS per (chunk, tap), two tiles deep like the GEMM engine.  Vector-memory traffic per
// launch, C = 64 / 128 / 256 at B = 8, 800 x 1333: 1230 -> 270, 1230 -> 230, 620 -> 240 MB.
// The dgrad form reads dy through the same patch with the taps flipped and the kernel transposed (K = Co).
#pragma once
#include "gemm_bf16_core.h"

namespace detr {

// Tile: TH x 32 output pixels.  TH = 8 (256 pixels, 4 MFMA row blocks per wave, 2 workgroups per CU) or TH = 4 (128 pixels, 2 row
// blocks, 33 + 20 KB of LDS and <= 168 VGPRs: 3 workgroups per CU; half the pixels per kernel-tile fetch, twice the workgroups).
constexpr int CH_TW = 32;                                // output pixels of a tile row = one MFMA row block
constexpr int CH_PW = CH_TW + 2;                         // input patch width (1 pixel border)
template <int TH>
struct HaloGeom {
    static constexpr int PH = TH + 2;                    // patch rows
    static constexpr int PIX = PH * CH_PW;               // 340 / 204 patch pixels
    static constexpr int GRAN = PIX * 4;                 // 16-byte granules (8 channels) of a 32-channel chunk
    static constexpr int NG = (GRAN + 255) / 256;        // granules per thread (the last round is partial)
    static constexpr int RB = TH / 2;                    // 32-pixel row blocks per wave
};

template <int BN, int TH>
struct ConvHaloSmem {
    unsigned short P[2][HaloGeom<TH>::PIX][BF_LD];       // input patch of chunk c in P[c & 1]
    unsigned short B[2][BN][BF_LD];                      // kernel tile of step s in B[s & 1]
};
template <int BN, int TH>
struct ConvHaloSmemBytes {
    static constexpr int TILES = (int)sizeof(ConvHaloSmem<BN, TH>);
    static constexpr int STAGE = StageCfg<BN, 2>::BYTES;
    static constexpr int VALUE = TILES > STAGE ? TILES : STAGE;
};

// Epilogue of the 256-pixel tile: y = mask(relu(acc + bias)) rounded to bf16.  One strip = one tile row (32 pixels) x the
// wave's BN / 2 channels, transposed through a WAVE-PRIVATE LDS region (no workgroup barrier between strips); a lane owns
// 8 consecutive channels of a pixel: one 16-byte mask load, one 16-byte store, all of a strip's mask loads requested
// before its LDS round trip.  (The generic epilogue of gemm_core.h walks a strip in a rolled loop with the bias / mask
// loads inside: with 4 strips per wave and only 8 waves per CU that was 16 dependent L2 round trips per workgroup --
// 71 of the 103 us of the C = 64 launch, ablation build 31.)
// S2C (stride-2 input-gradient class, see the kernel): tile coordinates are CLASS coordinates (h2, w2); the pixel written (and the
// mask pixel read) is (2 h2 + ph, 2 w2 + pw) of the [Hd][Wd] tensor.
template <int BN, int TH, bool S2C = false>
__device__ __forceinline__ void halo_epilogue(const f32x16 (&acc)[TH / 2][BN / 64], float *stage_base, unsigned short *dst,
                                              const ConvArgs &a, int n, int h0, int w0, int n0, int wm, int wn, int lane,
                                              int wave, int ph = 0, int pw = 0) {
    constexpr int TN = BN / 64, WTN = BN / 2;
    constexpr int LD = WTN + 4;                         // floats per staged row (16-byte aligned rows)
    constexpr int L8 = WTN / 8;                         // lanes per pixel
    constexpr int RPI = 64 / L8, ITERS = 32 / RPI;      // pixels per pass, passes per strip
    float *stage = stage_base + wave * (32 * LD);
    const int l31 = lane & 31, rh = (lane >> 5) * 4;
    const int c8 = (lane % L8) * 8, rsub = lane / L8;
    const int col = n0 + wn * WTN + c8;
    const EpiArgs &e = a.e;
    float bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bi[j] = 0.0f;
    if (e.bias) {
        const float4 b0 = *reinterpret_cast<const float4 *>(e.bias + col), b1 = *reinterpret_cast<const float4 *>(e.bias + col + 4);
        bi[0] = b0.x; bi[1] = b0.y; bi[2] = b0.z; bi[3] = b0.w; bi[4] = b1.x; bi[5] = b1.y; bi[6] = b1.z; bi[7] = b1.w;
    }
    const unsigned short *mask = reinterpret_cast<const unsigned short *>(e.mask);
    const int PS = S2C ? 2 : 1;                         // pixel step of the output tensor per tile pixel
    const int wn_ok = S2C ? (a.Wd - pw + 1) / 2 - w0 : a.Wd - w0;        // valid pixels of a tile row
#pragma unroll
    for (int mi = 0; mi < TH / 2; ++mi) {
        const int h = S2C ? 2 * (h0 + (TH / 2) * wm + mi) + ph : h0 + (TH / 2) * wm + mi;
        const bool row_ok = h < a.Hd;
        const long long prow0 = ((long long)n * a.Hd + h) * a.Wd + (S2C ? 2 * w0 + pw : w0);
        uint4 mk[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int tw = it * RPI + rsub;
            mk[it] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
            if (mask && row_ok && tw < wn_ok) {
                if (e.m16 == 2) mk[it].x = reinterpret_cast<const unsigned char *>(e.mask)[(prow0 + PS * tw) * e.ldmask + (col >> 3)];   // 8 mask bits
                else mk[it] = epi_ld16(mask + (prow0 + PS * tw) * e.ldmask + col);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();                // the previous strip's reads are done
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + rh) * LD + ni * 32 + l31] = acc[mi][ni][r];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int tw = it * RPI + rsub;
            const float4 v0 = *reinterpret_cast<const float4 *>(stage + tw * LD + c8);
            const float4 v1 = *reinterpret_cast<const float4 *>(stage + tw * LD + c8 + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            unsigned mw[4] = {mk[it].x, mk[it].y, mk[it].z, mk[it].w};
            if (mask && e.m16 == 2) {                  // bit-packed mask: expand the byte into 1.0 / 0.0 bf16 pairs
                const unsigned mb = mk[it].x;
#pragma unroll
                for (int j = 0; j < 4; ++j) mw[j] = ((mb >> (2 * j)) & 1u ? 0x3f80u : 0u) | ((mb >> (2 * j + 1)) & 1u ? 0x3f800000u : 0u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] += bi[j];
                if (e.act == 1) v[j] = fmaxf(v[j], 0.0f);
                const float m = bf16_bits_to_f32((j & 1) ? (mw[j >> 1] >> 16) : (mw[j >> 1] & 0xFFFFu));
                v[j] = (m > 0.0f) ? v[j] : 0.0f;
            }
            if (row_ok && tw < wn_ok) {
                const uint4 ow = make_uint4(f32_to_bf16_pair(v[0], v[1]), f32_to_bf16_pair(v[2], v[3]), f32_to_bf16_pair(v[4], v[5]), f32_to_bf16_pair(v[6], v[7]));
                *reinterpret_cast<uint4 *>(dst + (prow0 + PS * tw) * a.Cd + col) = ow;
                if (e.mbits_out) e.mbits_out[(prow0 + PS * tw) * e.ld_mbits_out + (col >> 3)] = (unsigned char)bf16x8_gt0_bits(ow.x, ow.y, ow.z, ow.w);
            }
        }
    }
}

// Kernel tile of the input-gradient form ([n = ci][k = co], k contiguous, 32 k = four 16-byte chunks per row): UNPADDED 64-byte rows with
// the chunk index XOR-swizzled by (row >> 2) & 3.  The padded 80-byte rows of the GEMM engine make the fragment reads conflict-free but
// not the stores (the 8 lanes of a ds_write_b128 group cover two rows whose first slots share a bank: SQ_LDS_BANK_CONFLICT 0.12 of the CU
// cycles in the dgrad launches against 0.04 forward); with the swizzle both are conflict-free (enumerated against the bank model of
// MI355X_MICROARCH.md: ds_read_b128 groups of 16 lanes over 64 banks, ds_write_b128 groups of 8 lanes over 32).
__device__ __forceinline__ int halo_bsw(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3); }      // in shorts
template <int BN>
__device__ __forceinline__ void halo_store_b_swz(unsigned short *flat, const uint4 (&r)[BN / 64], int tid) {
#pragma unroll
    for (int i = 0; i < BN / 64; ++i) *reinterpret_cast<uint4 *>(flat + halo_bsw((tid >> 2) + 64 * i, tid & 3)) = r[i];
}

// ConvArgs as for conv3x3_bf16c_kernel (stride 1, pad 1, bf16 src / w / dst); Hp / Wp carry the tile counts along H / W.
// S2C: the four pixel-parity classes (ph, pw) of the STRIDE-2 input gradient in one launch (ConvArgs.cls_*).  dx[2 h2 + ph, 2 w2 + pw] only receives the taps kh = ph + 1
// (mod 2), kw = pw + 1 (mod 2), reading dy[h2 + dh, w2 + dw] with dh, dw in {0, 1}: a stride-1 "2 x 2" convolution over dy in class
// coordinates.  The kernel walks FOUR taps per chunk (patch offsets (1..2, 1..2) of the same haloed dy patch) for every class; the taps
// a class does not have request no kernel tile (out-of-range offsets) and skip their MFMAs -- wave-uniform, the pipeline keeps its
// static shape.  (The tile kernel ran the four class launches of a conv at 220-330 TFLOP/s: gathered operands, 1-4 short tap-GEMMs.)
template <int BN, bool DGRAD, int TH = 8, bool S2C = false>
__global__ __launch_bounds__(GEMM_THREADS, TH == 8 ? 2 : 3) void conv3x3_halo_bf16_kernel(ConvArgs a) {
    static_assert(!S2C || DGRAD, "the class form exists for the input gradient only");
    using T = TileCfg<TH * CH_TW, BN, 2, 2>;             // wave tile: TH / 2 pixel rows x BN / 2 channels
    using G = HaloGeom<TH>;
    constexpr int CH_PIX = G::PIX, CH_GRAN = G::GRAN, CH_NG = G::NG;
    extern __shared__ __attribute__((aligned(16))) char halo_smem[];
    ConvHaloSmem<BN, TH> &sm = *reinterpret_cast<ConvHaloSmem<BN, TH> *>(halo_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    int ph = 0, pw = 0, Hp = a.Hp, Wp = a.Wp;
    if constexpr (S2C) {                                 // the four classes share the launch: [cls_off[k], cls_off[k + 1])
        int k;
        if (a.cls_mix) {
            // Grid order: the four classes ROUND-ROBIN (workgroup id -> class id & 3), the classes' surplus tiles behind them.  A class writes (and reads the mask
            // of) the pixels of ONE column parity only -- at 128 channels that is address bit 8, every other 256-byte block of the tensor, for the whole lifetime
            // of its workgroups -- so with one class after the other (round 3's "heaviest first") the launch used half of the memory channels at a time:
            // 120.2 / 94.1 / 113.9 us at 128 / 256 / 512 channels against 115.5 / 81.7 / 99.6 round-robin (profiles/r05_ab_results.txt #9).
            const int nk[4] = {a.cls_off[1], a.cls_off[2] - a.cls_off[1], a.cls_off[3] - a.cls_off[2], a.cls_off[4] - a.cls_off[3]};
            int m = nk[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) m = nk[q] < m ? nk[q] : m;
            if (id < 4 * m) { k = id & 3; id >>= 2; }
            else {
                int j = id - 4 * m;
                k = 0;
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (k == q && j >= nk[q] - m) { j -= nk[q] - m; k = q + 1; }
                id = m + j;
            }
        } else {
            k = (id >= a.cls_off[1]) + (id >= a.cls_off[2]) + (id >= a.cls_off[3]);
            id -= a.cls_off[k];
        }
        ph = (k == 0 || k == 2) ? 1 : 0;
        pw = (k == 0 || k == 1) ? 1 : 0;
        Hp = a.cls_hp[k];
        Wp = a.cls_wp[k];
    }
    const int tn = id % a.tiles_n;
    int t = id / a.tiles_n;
    const int twi = t % Wp;
    t /= Wp;
    const int thi = t % Hp;
    const int n = t / Hp;
    const int h0 = thi * TH, w0 = twi * CH_TW, n0 = tn * BN;
    const int nchunks = a.Cs / BF_BK;
    const long long tapstride = (long long)a.Ci * a.Co;

    // ---- patch loader: granule g = tid + 256 * i -> patch pixel g >> 2, channels 8 * (g & 3) of the chunk
    BufSrc src;
    src.init_bytes(a.src, (long long)a.N * a.Hs * a.Ws * a.Cs * 2);
    unsigned goff[CH_NG];
#pragma unroll
    for (int i = 0; i < CH_NG; ++i) {
        const int g = tid + 256 * i;
        const int px = g >> 2;
        const int pr = px / CH_PW, pc = px - pr * CH_PW;
        const int h = h0 - 1 + pr, w = w0 - 1 + pc;
        const bool ok = g < CH_GRAN && h >= 0 && h < a.Hs && w >= 0 && w < a.Ws;
        goff[i] = ok ? ((unsigned)((n * a.Hs + h) * a.Ws + w) * (unsigned)a.Cs + (unsigned)(g & 3) * 8u) * 2u : BUF_OOB;
    }
    auto patch_load = [&](int c, uint4 (&r)[CH_NG]) {      // chunk c >= nchunks: every lane out of range, no traffic
        const bool live = c < nchunks;
#pragma unroll
        for (int i = 0; i < CH_NG; ++i) r[i] = src.ld16((live && goff[i] != BUF_OOB) ? goff[i] + (unsigned)c * (BF_BK * 2u) : BUF_OOB);
    };
    auto patch_store = [&](int buf, const uint4 (&r)[CH_NG]) {
        unsigned short *flat = &sm.P[buf][0][0];
#pragma unroll
        for (int i = 0; i < CH_NG; ++i) {
            const int g = tid + 256 * i;
            if (i + 1 < CH_NG || g < CH_GRAN)
                *reinterpret_cast<uint4 *>(flat + (g >> 2) * BF_LD + (g & 3) * 8) = r[i];
        }
    };
    // ---- kernel loader: fwd [k = ci][n = co] (transpose-read image), dgrad [n = ci][k = co]
    using LB = typename std::conditional<DGRAD, LoaderKh<BN>, LoaderMNth<BN, true>>::type;
    constexpr int NRB = LB::NREG;
    LB lb;
    lb.init(a.w, a.Co, n0, a.Cd, a.Cs, true, tid, 9 * tapstride);
    // S2C: step tp in 0..3 = patch offset (1 + (tp >> 1), 1 + (tp & 1)); the class has the tap iff (offset 1 or parity 1) on each axis;
    // offset 1 reads dy[h2] (kernel row 1 for parity 0, row 2 for parity 1), offset 2 reads dy[h2 + 1] (kernel row 0)
    auto s2_tap_ok = [&](int tp) -> bool { return ((tp >> 1) == 0 || ph == 1) && ((tp & 1) == 0 || pw == 1); };
    auto s2_tap_w = [&](int tp) -> int {
        const int kh = (tp >> 1) == 0 ? (ph ? 2 : 1) : 0, kw = (tp & 1) == 0 ? (pw ? 2 : 1) : 0;
        return kh * 3 + kw;
    };
    auto b_load = [&](int c, int tp, typename LB::Reg (&rb)[NRB]) {      // step (chunk c, patch offset tp = 3 * dh + dw)
        const bool live = c < nchunks && (!S2C || s2_tap_ok(tp));
        const int wt = S2C ? s2_tap_w(tp) : (DGRAD ? 8 - tp : tp);        // dgrad: the kernel tap is the flipped offset
        lb.load(c * BF_BK, live ? a.Cs : 0, rb, live ? (unsigned)(wt * tapstride * 2) : 0u);
    };

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int l31 = lane & 31, kh8 = (lane >> 5) * 8;
    // A fragment of tile row (RB * wm + mi), patch offset (dh, dw), k-step ks: patch row ((RB * wm + mi + dh) * 34 + l31 + dw)
    const unsigned short *pa_base = &sm.P[0][0][0] + ((G::RB * wm) * CH_PW + l31) * BF_LD + kh8;
    auto mma_tap = [&](int pbuf, int bbuf, int tp) {
        const int dh = S2C ? 1 + (tp >> 1) : tp / 3, dw = S2C ? 1 + (tp & 1) : tp - 3 * (tp / 3);
        const unsigned short *pa = pa_base + pbuf * (CH_PIX * BF_LD) + (dh * CH_PW + dw) * BF_LD;
        const unsigned short(*Bs)[BF_LD] = sm.B[bbuf];
#pragma unroll
        for (int ks = 0; ks < BF_BK; ks += 16) {
            bf16x8 fa[T::TM], fb[T::TN];
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi) {
                if constexpr ((DETR_ABLATE & 16) != 0) fa[mi] = __builtin_bit_cast(bf16x8, make_uint4(lane, mi, ks, 0x3f803f80u));
                else fa[mi] = *reinterpret_cast<const bf16x8 *>(pa + mi * (CH_PW * BF_LD) + ks);
            }
#pragma unroll
            for (int ni = 0; ni < T::TN; ++ni) {
                if constexpr ((DETR_ABLATE & 16) != 0) fb[ni] = __builtin_bit_cast(bf16x8, make_uint4(lane, ni, ks, 0x3f803f80u));
                else if (!DGRAD) fb[ni] = frag_tr<BN>(Bs, wn * T::WTN + ni * 32, ks, lane);
                else fb[ni] = *reinterpret_cast<const bf16x8 *>(&Bs[0][0] + halo_bsw(wn * T::WTN + ni * 32 + l31, (ks + kh8) >> 3));
            }
#pragma unroll
            for (int mi = 0; mi < T::TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::TN; ++ni) {
                    if constexpr ((DETR_ABLATE & 1) != 0) { ablate_keep(fa[mi]); ablate_keep(fb[ni]); }
                    else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
                }
        }
    };

    // ---- pipeline.  Step s = 9 * c + tp multiplies patch P[c & 1] at offset tp with kernel tile B[s & 1].  The kernel tile
    // of step s + 1 waits in one register set (stored at the top of step s), the tile of step s + 2 is in flight into the
    // other; the patch of chunk c + 1 is requested at tp = 4 of chunk c - 1 and stored at tp = 2 of chunk c (its buffer was
    // last read in chunk c - 1).  Every request and LDS store is unconditional (past the end: out-of-range offsets, a buffer
    // nobody reads) and the nine taps are unrolled, so that the compiler's vmcnt waits are exact -- see gemm_bf16c_body.
    uint4 rp[CH_NG];
    typename LB::Reg rb0[NRB], rb1[NRB];
    patch_load(0, rp);
    b_load(0, 0, rb0);
    patch_store(0, rp);
    if constexpr (DGRAD) halo_store_b_swz<BN>(&sm.B[0][0][0], rb0, tid); else lb.store(sm.B[0], rb0);
    patch_load(1, rp);
    b_load(0, 1, rb0);
    b_load(0, 2, rb1);
    lds_barrier();
    // one chunk; PAR = parity of its first step (9 taps per chunk: chunks alternate)
    auto chunk = [&](const int c, auto par) {
        constexpr int PAR = decltype(par)::value;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int cur = (PAR + tp) & 1;
            // kernel tile of the next step: register set `cur ^ PAR ...` -- step parity decides the set, statically
            if (((PAR + tp) & 1) == 0) {
                if constexpr ((DETR_ABLATE & 4) == 0) { if constexpr (DGRAD) halo_store_b_swz<BN>(&sm.B[cur ^ 1][0][0], rb0, tid); else lb.store(sm.B[cur ^ 1], rb0); }
                else for (int i = 0; i < NRB; ++i) ablate_keep(rb0[i]);
                if constexpr ((DETR_ABLATE & 2) == 0) b_load(c + (tp + 3) / 9, (tp + 3) % 9, rb0);
            } else {
                if constexpr ((DETR_ABLATE & 4) == 0) { if constexpr (DGRAD) halo_store_b_swz<BN>(&sm.B[cur ^ 1][0][0], rb1, tid); else lb.store(sm.B[cur ^ 1], rb1); }
                else for (int i = 0; i < NRB; ++i) ablate_keep(rb1[i]);
                if constexpr ((DETR_ABLATE & 2) == 0) b_load(c + (tp + 3) / 9, (tp + 3) % 9, rb1);
            }
            if (tp == 2) {
                if constexpr ((DETR_ABLATE & 4) == 0) patch_store((c + 1) & 1, rp);
                else for (int i = 0; i < CH_NG; ++i) ablate_keep(rp[i]);
            }
            if (tp == 4) {
                if constexpr ((DETR_ABLATE & 2) == 0) patch_load(c + 2, rp);
            }
            mma_tap(c & 1, cur, tp);
            if constexpr ((DETR_ABLATE & 8) == 0) lds_barrier();
        }
    };
    // S2C: four steps per chunk (step parity = tp & 1 in every chunk); the patch of chunk c + 1 -- requested one chunk ago -- is stored at
    // the top of chunk c (its buffer was last read in chunk c - 1) and the patch of chunk c + 2 is requested right behind it
    auto chunk4 = [&](const int c) {
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const int cur = tp & 1;
            if ((tp & 1) == 0) {
                if constexpr (DGRAD) halo_store_b_swz<BN>(&sm.B[cur ^ 1][0][0], rb0, tid);        // (the class form is an input gradient)
                else lb.store(sm.B[cur ^ 1], rb0);
                b_load(c + (tp + 3) / 4, (tp + 3) % 4, rb0);
            } else {
                if constexpr (DGRAD) halo_store_b_swz<BN>(&sm.B[cur ^ 1][0][0], rb1, tid);
                else lb.store(sm.B[cur ^ 1], rb1);
                b_load(c + (tp + 3) / 4, (tp + 3) % 4, rb1);
            }
            if (tp == 0) {
                patch_store((c + 1) & 1, rp);
                patch_load(c + 2, rp);
            }
            if (s2_tap_ok(tp)) mma_tap(c & 1, cur, tp);
            lds_barrier();
        }
    };
    if constexpr (S2C) {
        for (int c = 0; c < nchunks; ++c) chunk4(c);
    } else {
        int c = 0;
        for (; c + 2 <= nchunks; c += 2) {
            chunk(c, std::integral_constant<int, 0>{});
            chunk(c + 1, std::integral_constant<int, 1>{});
        }
        if (c < nchunks) chunk(c, std::integral_constant<int, 0>{});
    }
    __syncthreads();
    halo_epilogue<BN, TH, S2C>(acc, reinterpret_cast<float *>(halo_smem), reinterpret_cast<unsigned short *>(a.dst), a, n, h0, w0, n0, wm, wn, lane, wave, ph, pw);
}

template <int BN, int TH>
static int launch_conv_halo_t(const ConvArgs &a0, bool dgrad, hipStream_t s) {
    ConvArgs a = a0;
    a.Hp = cdiv(a.Hd, TH);
    a.Wp = cdiv(a.Wd, CH_TW);
    a.tiles_m = a.N * a.Hp * a.Wp;
    a.tiles_n = cdiv(a.Cd, BN);
    constexpr int smem = ConvHaloSmemBytes<BN, TH>::VALUE;
    static bool reserved[2] = {false, false};
    const void *fn = dgrad ? reinterpret_cast<const void *>(conv3x3_halo_bf16_kernel<BN, true, TH>)
                           : reinterpret_cast<const void *>(conv3x3_halo_bf16_kernel<BN, false, TH>);
    if (!reserved[dgrad ? 1 : 0]) {
        hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        DETR_REQUIRE(err == hipSuccess, "conv3x3 (halo): cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(err));
        reserved[dgrad ? 1 : 0] = true;
    }
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(GEMM_THREADS);
    if (dgrad) hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<BN, true, TH>), grid, block, smem, s, a);
    else hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<BN, false, TH>), grid, block, smem, s, a);
    return 0;
}

// the four pixel-parity classes of a stride-2 input gradient in one launch (4-row tiles; heaviest class first)
template <int BN>
static int launch_conv_halo_s2classes(const ConvArgs &a0, hipStream_t s) {
    constexpr int TH = 4;
    ConvArgs a = a0;
    a.tiles_n = cdiv(a.Cd, BN);
    a.cls_off[0] = 0;
    for (int k = 0; k < 4; ++k) {
        const int ph = (k == 0 || k == 2) ? 1 : 0, pw = (k == 0 || k == 1) ? 1 : 0;
        const int Hc = (a.Hd - ph + 1) / 2, Wc = (a.Wd - pw + 1) / 2;        // pixels of the class along H / W
        a.cls_hp[k] = cdiv(Hc, TH);
        a.cls_wp[k] = cdiv(Wc, CH_TW);
        a.cls_off[k + 1] = a.cls_off[k] + a.N * a.cls_hp[k] * a.cls_wp[k] * a.tiles_n;
    }
    if (a.cls_off[4] <= 0) return 0;
    a.cls_mix = tune(T_DGRAD_S2_CLASSES) == 4 ? 0 : 1;       // DETR_HIP_DGRAD_S2_CLASSES = 4: one class after the other (A/B of the grid order)
    constexpr int smem = ConvHaloSmemBytes<BN, TH>::VALUE;
    static bool reserved = false;
    const void *fn = reinterpret_cast<const void *>(conv3x3_halo_bf16_kernel<BN, true, TH, true>);
    if (!reserved) {
        hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        DETR_REQUIRE(err == hipSuccess, "conv3x3 (halo, stride-2 classes): cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(err));
        reserved = true;
    }
    hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<BN, true, TH, true>), dim3((unsigned)a.cls_off[4]), dim3(GEMM_THREADS), smem, s, a);
    return 0;
}

// Tile height.  Measured (scripts/micro_conv.py, profiles/r03_micro_conv_halo_tiles.txt): 4-row tiles win 10-13 % on the 128- and
// 256-channel maps (100x167: 58.7 -> 52.0 us forward, 50x84: 61.6 -> 53.4; three workgroups per CU, 4 % instead of 12 % padded rows on
// the 50-row map, 624 instead of 336 workgroups) and lose on the 64-channel forward (64.3 -> 69.9 us: HBM / epilogue bound, twice the
// kernel-tile fetches).  DETR_HIP_CONV_HALO = 3 forces 4-row tiles, 4 forces 8-row tiles; 0 = this rule.
template <int BN>
static int launch_conv_halo(const ConvArgs &a, bool dgrad, hipStream_t s) {
    const int mode = tune(T_CONV_HALO);
    const bool th4 = mode == 3 || (mode != 4 && a.Cd >= 128);
    return th4 ? launch_conv_halo_t<BN, 4>(a, dgrad, s) : launch_conv_halo_t<BN, 8>(a, dgrad, s);
}

}  // namespace detr
