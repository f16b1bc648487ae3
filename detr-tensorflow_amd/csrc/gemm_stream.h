// gemm_stream.h -- streaming short-K GEMM for the bf16-stored backbone activations.
//
// The 1x1 convolutions of ResNet layer1 / layer2 (resnet_backbone.py:119-135 and their input gradients) are
// C[M, N] = A[M, K] * B with M = B*H*W in the hundreds of thousands and K = 64 .. 256: 1-4 MFMAs per 32x32 output
// tile against an epilogue that reads the residual (and the ReLU mask) and writes C.  They are HBM streams, and the
// generic tile engine (gemm_f32.hip) runs them at ~3 TB/s: per workgroup one operand round trip, two barriers and a
// rolled four-item epilogue with a round trip per item, ~0.45 instructions per output element.  A plain elementwise
// kernel moves the same traffic mix at 5.5-6 TB/s on this chip (scripts/experiments/hbm_mix_probe.py).
//
// This kernel is built like the elementwise kernel instead:
//   * persistent waves: a workgroup stages its 64-column slice of B (K x 64 bf16, <= 17 KB) in LDS once and its four
//     waves then walk independently over 32-row strips of A -- no workgroup barrier after the prologue;
//   * the MFMA is issued with the operands swapped (D^T = B^T A^T), so (a) the A fragment of a lane is 16 contiguous
//     bytes of one row of A in global memory and is loaded straight into registers, one strip ahead, and (b) every
//     accumulator quad holds 4 consecutive columns of one row: the wave-private LDS transposition is 8 ds_write_b128
//     + 8 ds_read_b128 per 32x64 strip and leaves each lane with 8 consecutive columns (16 bytes of bf16);
//   * residual, mask and output are 16-byte accesses covering whole 128-byte lines (8 rows per wave instruction); the
//     residual / mask of a strip are requested before its MFMAs, the A rows of the next strip before its epilogue;
//   * the workgroups that share A rows (the N / 64 column slices) are placed on the same XCD so that A is read from
//     HBM once and from that XCD's L2 otherwise.
// Arithmetic: bf16 x bf16 -> fp32 MFMA (v_mfma_f32_32x32x16_bf16), epilogue in fp32 in the order of gemm_core.h
// epi_one (bias, residual, ReLU, mask), one RNE rounding to bf16.
#pragma once
#include "gemm_core.h"
#include "gemm_bf16_core.h"

namespace detr {

struct StreamArgs {
    int M, N;
    const unsigned short *A; long long lda;        // [M][lda] bf16, K contiguous
    const unsigned short *B; long long ldb;        // BKC: [N][ldb] (K contiguous); else [K][ldb] (N contiguous)
    unsigned short *C; long long ldc;              // [M][ldc] bf16
    const unsigned short *res; long long ldr;      // optional [M][ldr] bf16
    const unsigned short *mask; long long ldm;     // optional [M][ldm] bf16, keeps C where mask > 0; MASK == 2: bytes of 8 mask bits, ldm in bytes
    unsigned char *mbits_out; long long ld_mbits;  // optional: also write (C > 0) as one byte per 8 outputs (gemm_core.h EpiArgs)
    const float *bias;                             // optional [N] fp32
    int act;                                       // 0 none, 1 ReLU
    int n_tiles;                                   // N / 64
    int row_tiles;                                 // cdiv(M, 32)
    int q;                                         // workgroups per (XCD, column slice); grid = 8 * n_tiles * q
    // EXT instantiations only (the transformer's K = 256 FFN GEMMs): v = (acc + bias) * alpha, keyed dropout after the
    // activation (no residual) or before the residual add -- the positions of gemm_core.h epi_one
    float alpha;
    float drop_scale;                              // 1 / (1 - p), 0 = no dropout
    uint32_t drop_thresh, drop_seed;
    const uint32_t *drop_step;
};

// Cache policy of the epilogue streams.  The residual and the mask are read exactly once by this kernel and by nobody
// after it, so they are requested non-temporal (bit 0): in the step that is -0.2 ms (20.07 -> 19.85 ms), e.g. M534400 N256
// K64 +res+mask 166 -> 146 us, because they stop evicting the A rows the other slices of the workgroup are about to re-read.
// The output stores stay temporal: the next kernel reads C, and non-temporal stores measured +-0 here and +0.02 ms on the consumers.
#ifndef DETR_STREAM_NT
#define DETR_STREAM_NT 1
#endif
__device__ __forceinline__ uint4 stream_ld_ep(const BufSrc &src, unsigned off) {
#if DETR_STREAM_NT & 1
    return src.ld16_nt(off);
#else
    return src.ld16(off);
#endif
}
#ifndef DETR_STREAM_PIPE
#define DETR_STREAM_PIPE 1                         // 0: the compiler's own order of fragment reads and MFMAs (A/B builds)
#endif
#ifndef DETR_STREAM_PROF
#define DETR_STREAM_PROF 0                         // 1 (probe builds, scripts/experiments/stream_probe.py): s_memtime per strip section into maskbits_out
#endif
#if DETR_STREAM_PROF
#define STREAM_TICK(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[k] += t_ - prof_last; prof_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STREAM_TICK(k) do { } while (0)
#endif
constexpr int STREAM_LD = 68;                      // floats per staged row (64 + 4): conflict-free b128 writes and reads

// SL = column slices (64 columns each) a workgroup owns: its 4 waves are SL slice owners x 4 / SL row walkers, and the SL
// owners of one walker work on the SAME 32 rows at the same time, so the A rows of a strip leave L2 once per workgroup
// instead of once per slice.  (Measured: the kernel's time is proportional to the number of slice-strips -- M534400 K64:
// 33 us for N = 64 at 6.2 TB/s, 131 us for N = 256 -- i.e. the A rows re-read from L2 by the other slices cost as much as
// HBM bytes; scripts/experiments/stream_stride_probe.py.)
// NW = column slices per WAVE (round 4, K = 256 only): a wave multiplies the A rows it holds with two resident slices (four MFMAs per
// A fragment instead of two) -- the row-per-lane A requests (32 rows x 32 bytes per wave instruction, one line lookup per row) are what a
// K = 256 launch with 16-32 column slices spends a third of its time on (profiles/r04_ab_results.txt #15) -- in workgroups of 8 waves
// (2 x 34 KB of B + 8 strips of staging = 137 KB: one workgroup per CU, the same 8 waves per CU as two 4-wave workgroups before).
template <int K, int SL, int NW = 1>
struct StreamSmem {
    static constexpr int WAVES = (NW > 1 && K * NW * SL >= 512) ? 8 : 4;      // 8 waves where four strips of staging would leave one 4-wave workgroup per CU
    unsigned short B[SL * NW][64][K + 8];          // [slice][n][k], +8 bf16 of padding: 16-byte fragment reads spread over the banks
    float stage[WAVES][32][STREAM_LD];             // one 32 x 64 fp32 strip per wave
};

__device__ __forceinline__ void stream_unpack8(uint4 r, float (&o)[8]) {
    o[0] = bf16_bits_to_f32(r.x & 0xFFFFu); o[1] = __builtin_bit_cast(float, r.x & 0xFFFF0000u);
    o[2] = bf16_bits_to_f32(r.y & 0xFFFFu); o[3] = __builtin_bit_cast(float, r.y & 0xFFFF0000u);
    o[4] = bf16_bits_to_f32(r.z & 0xFFFFu); o[5] = __builtin_bit_cast(float, r.z & 0xFFFF0000u);
    o[6] = bf16_bits_to_f32(r.w & 0xFFFFu); o[7] = __builtin_bit_cast(float, r.w & 0xFFFF0000u);
}

// workgroups per CU that fit the 160 KB of LDS (at most 3: 12 waves / CU already keep > 100 KB of requests in flight)
template <int K, int SL, int NW = 1>
struct StreamOcc {
    static constexpr int BYTES = (int)sizeof(StreamSmem<K, SL, NW>);
    static constexpr int VALUE = (3 * BYTES <= 160 * 1024) ? 3 : ((2 * BYTES <= 160 * 1024) ? 2 : 1);
};

// MASK: 0 none, 1 a bf16 tensor, 2 bit-packed (one byte per 8 columns)
// MBO: the epilogue also writes (C > 0) as bits (StreamArgs.mbits_out; instantiated for the forward form only: [k][n] weights + residual)
template <int K, bool BKC, bool RES, int MASK, int SL = 1, bool EXT = false, int NW = 1, bool MBO = false>
__global__ __launch_bounds__(64 * (StreamSmem<K, SL, NW>::WAVES), (StreamOcc<K, SL, NW>::VALUE)) void gemm_stream_bf16_kernel(StreamArgs a) {
    constexpr int WAVES = StreamSmem<K, SL, NW>::WAVES, THREADS = 64 * WAVES;
    constexpr int WPS = WAVES / SL;                // row walkers (waves per slice group) of a workgroup
    constexpr int KC = (K > 128) ? 128 : K;       // A rows are held in registers one K chunk (<= 128) at a time
    constexpr int NC = K / KC;                     // chunks per strip: 1, or an even number
    constexpr int KK = KC / 16;                    // MFMA k-steps per chunk
    static_assert(K % KC == 0 && (NC == 1 || NC % 2 == 0), "gemm_stream: K must be 64, 128 or a multiple of 256");
    __shared__ __attribute__((aligned(16))) StreamSmem<K, SL, NW> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup id -> (XCD, column slice, walker): ids are dealt round-robin over the 8 XCDs, so the n_tiles slices of
    // one walker p are consecutive multiples of 8 apart -- same XCD, dispatched together, same A rows
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int nt = j % a.n_tiles, p = (j / a.n_tiles) * 8 + xcd;      // n_tiles = slice GROUPS of SL slices
    const int ws = wave % SL, wr = wave / SL;      // this wave's slice inside the group, its row walker
    const int n0 = (nt * SL + ws) * NW * 64;       // first column of this wave's NW slices
    const int stride = a.q * 8 * WPS;              // wave slots per column slice

    // ---- per-lane constants ----------------------------------------------------------------------------------------------
    const int l31 = lane & 31, h = lane >> 5;
    const int erow = lane >> 3, ecg = lane & 7;    // epilogue item: 8 rows x 8 column groups of 8
    float bias[NW][8];
#pragma unroll
    for (int sw = 0; sw < NW; ++sw)
#pragma unroll
        for (int i = 0; i < 8; ++i) bias[sw][i] = a.bias ? a.bias[n0 + sw * 64 + ecg * 8 + i] : 0.0f;
    BufSrc srcA, srcR, srcM;
    srcA.init_bytes(a.A, ((long long)(a.M - 1) * a.lda + K) * 2);
    if (RES) srcR.init_bytes(a.res, ((long long)(a.M - 1) * a.ldr + a.N) * 2);
    if (MASK == 1) srcM.init_bytes(a.mask, ((long long)(a.M - 1) * a.ldm + a.N) * 2);
    if (MASK == 2) srcM.init_bytes(a.mask, (long long)(a.M - 1) * a.ldm + a.N / 8);
    BufSrc dstC, dstB;
    dstC.init_bytes(a.C, ((long long)(a.M - 1) * a.ldc + a.N) * 2);
    if (MBO) dstB.init_bytes(a.mbits_out, (long long)(a.M - 1) * a.ld_mbits + a.N / 8);
    const unsigned ldc2 = (unsigned)(a.ldc * 2), ldr2 = (unsigned)(a.ldr * 2), ldm2 = (unsigned)(a.ldm * 2), ldm1 = (unsigned)a.ldm, ldb1 = (unsigned)a.ld_mbits;
    const float act_floor = a.act == 1 ? 0.0f : -__builtin_huge_valf();      // ReLU without a branch: max(v, 0) or max(v, -inf)
    float *stage = &sm.stage[wave][0][0];
    const unsigned a_lane = (unsigned)(h * 16);    // byte offset of this lane's 8 k values inside a 16-k step
    const uint32_t dkey = (EXT && a.drop_scale != 0.0f) ? drop_key(a.drop_seed, a.drop_step) : 0u;

    auto load_a = [&](int rt, int chunk, uint4 (&f)[KK]) {
        const int row = rt * 32 + l31;
        const bool a_live = (DETR_ABLATE & 2) == 0 || rt < stride;      // ablation bit 1: only a wave's first strip is requested
        const unsigned base = (a_live && rt < a.row_tiles && row < a.M) ? (unsigned)((long long)row * a.lda * 2) + a_lane + chunk * (KC * 2) : BUF_OOB;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) f[kk] = srcA.ld16(base == BUF_OOB ? BUF_OOB : base + kk * 32);
    };
    // D^T[n][m] += sum_k B^T[n][k] A^T[k][m] over one chunk: lane&31 = m, accumulator r -> n = (r&3) + 8*(r>>2) + 4*h (+32*nh)
    // B fragment nq of k-step kk.  BKC: [n][k] rows (+8 bf16 of padding); else the transpose-read image of gemm_bf16_core.h ([4 k][16 n]
    // sub-blocks of the natural [k][n] orientation): lane (l31, h) receives the 8 consecutive k of column nh * 32 + l31 either way
    auto read_b = [&](int chunk, int kk, int nq) -> bf16x8 {
        const int nh = nq & 1, bs = ws * NW + (nq >> 1);
        if constexpr (BKC) return *reinterpret_cast<const bf16x8 *>(&sm.B[bs][nh * 32 + l31][chunk * KC + kk * 16 + h * 8]);
        else return frag_tr<64>(reinterpret_cast<const unsigned short(*)[8]>(&sm.B[bs][0][0]), nh * 32, chunk * KC + kk * 16, lane);
    };
    // D^T[n][m] += sum_k B^T[n][k] A^T[k][m] over one chunk: lane&31 = m, accumulator r -> n = (r&3) + 8*(r>>2) + 4*h (+32*nh).
    // The B fragments of k-step kk + 1 are read while the MFMAs of step kk run, one read per MFMA, pinned with sched_group_barrier
    // (round 4: left alone, the compiler put every fragment read right in front of its MFMA -- `ds_read, s_waitcnt, v_mfma` 64 times
    // per strip, an LDS round trip of latency per MFMA on waves that have one or two neighbours per SIMD to cover it)
    auto mma = [&](const uint4 (&f)[KK], int chunk, f32x16 (&acc)[2 * NW]) {
        constexpr int NQ = 2 * NW, RPF = BKC ? 1 : 2;      // LDS reads per fragment
        bf16x8 bq[2][NQ];
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) bq[0][nq] = read_b(chunk, 0, nq);
#if DETR_STREAM_PIPE
        if constexpr ((DETR_ABLATE & 1) == 0) sgb_ds_read<RPF * NQ>();
#endif
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, f[kk]);
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) {
                if (kk + 1 < KK) bq[(kk + 1) & 1][nq] = read_b(chunk, kk + 1, nq);
                if constexpr ((DETR_ABLATE & 1) != 0) { ablate_keep(bq[kk & 1][nq]); ablate_keep(af); }
                else acc[nq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[kk & 1][nq], af, acc[nq], 0, 0, 0);
            }
#if DETR_STREAM_PIPE
            if constexpr ((DETR_ABLATE & 1) == 0) {
#pragma unroll
                for (int nq = 0; nq < NQ; ++nq) {
                    if (kk + 1 < KK) sgb_ds_read<RPF>();
                    sgb_mfma();
                }
            }
#endif
        }
    };

#if DETR_STREAM_PROF
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_last = 0;
#endif
    // one strip: epilogue-operand requests, A prefetch, MFMAs, transposition, epilogue.  a_first holds chunk 0 of the
    // strip on entry.  NC == 1: the next strip's rows go to a_other and the two buffers swap roles from strip to strip
    // (the loop below is unrolled by two), so no register copy -- which would have to wait for the prefetch -- sits at
    // the end of a strip.  NC even: the chunks alternate between the buffers and chunk 0 of the next strip lands in
    // a_first again.
    auto strip = [&](const int rt, uint4 (&a_first)[KK], uint4 (&a_other)[KK]) {
        const int r0 = rt * 32;
        STREAM_TICK(7);                             // between strips (loop control)
        uint4 rres[NW][4], rmsk[NW][4];
#pragma unroll
        for (int sw = 0; sw < NW; ++sw)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = r0 + it * 8 + erow;
            const unsigned colb = (unsigned)((n0 + sw * 64 + ecg * 8) * 2);
            const bool ep_live = (DETR_ABLATE & 32) == 0;               // ablation bit 5: no residual / mask requests
            // (byte offsets fit 32 bits: the host dispatch only sends tensors below 4 GB here)
            if (RES) rres[sw][it] = stream_ld_ep(srcR, (ep_live && row < a.M) ? (unsigned)row * ldr2 + colb : BUF_OOB);
            if (MASK == 1) rmsk[sw][it] = stream_ld_ep(srcM, (ep_live && row < a.M) ? (unsigned)row * ldm2 + colb : BUF_OOB);
            if (MASK == 2)      // one byte = the 8 columns of this lane; the 8 lanes of a row read 8 consecutive bytes
                rmsk[sw][it].x = (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(srcM.rsrc, (ep_live && row < a.M) ? (unsigned)row * ldm1 + (unsigned)(((n0 + sw * 64) >> 3) + ecg) : BUF_OOB, 0, 0);
        }
        f32x16 acc[2 * NW];
#pragma unroll
        for (int nq = 0; nq < 2 * NW; ++nq)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nq][r] = 0.0f;
        STREAM_TICK(0);                             // epilogue-operand requests issued
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool last = (c + 1 == NC);
            if (c % 2 == 0) {
                load_a(last ? rt + stride : rt, last ? 0 : c + 1, a_other);
                mma(a_first, c, acc);
            } else {
                load_a(last ? rt + stride : rt, last ? 0 : c + 1, a_first);
                mma(a_other, c, acc);
            }
            STREAM_TICK(1 + (c & 1));               // chunk c: A requests + MFMAs issued (incl. the wait for its A rows)
        }
#pragma unroll
        for (int sw = 0; sw < NW; ++sw) {
            // ---- wave-private transposition: quads of 4 consecutive columns -> rows of 8 consecutive columns per lane
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(stage + l31 * STREAM_LD + nh * 32 + 8 * g + 4 * h) =
                        make_float4(acc[2 * sw + nh][4 * g], acc[2 * sw + nh][4 * g + 1], acc[2 * sw + nh][4 * g + 2], acc[2 * sw + nh][4 * g + 3]);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            STREAM_TICK(3);                         // accumulators staged (the first one waits for the MFMAs to finish)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = it * 8 + erow;
                const int row = r0 + rl;
                const float4 v0 = *reinterpret_cast<const float4 *>(stage + rl * STREAM_LD + ecg * 8);
                const float4 v1 = *reinterpret_cast<const float4 *>(stage + rl * STREAM_LD + ecg * 8 + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += bias[sw][i];
                bool keep[8];
                if constexpr (EXT) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= a.alpha;
                    if (a.drop_scale != 0.0f) {        // one hash per element pair (common.h); this lane's 8 columns start at an even index
                        const unsigned long long pair0 = ((unsigned long long)row * (unsigned long long)a.N + (unsigned)(n0 + sw * 64 + ecg * 8)) >> 1;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t hh = drop_hash(dkey, pair0 + j);
                            keep[2 * j] = (hh & 0xFFFFu) >= a.drop_thresh;
                            keep[2 * j + 1] = (hh >> 16) >= a.drop_thresh;
                        }
                        if (RES) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = keep[i] ? v[i] * a.drop_scale : 0.0f;
                        }
                    }
                }
                if (RES) {
                    float r[8];
                    stream_unpack8(rres[sw][it], r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += r[i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = vmax_raw(v[i], act_floor);
                if constexpr (EXT) {
                    if (!RES && a.drop_scale != 0.0f) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = keep[i] ? v[i] * a.drop_scale : 0.0f;
                    }
                }
                if (MASK == 1) {
                    float m[8];
                    stream_unpack8(rmsk[sw][it], m);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (m[i] > 0.0f) ? v[i] : 0.0f;
                }
                if (MASK == 2) {
                    const unsigned mb = rmsk[sw][it].x;
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = ((mb >> i) & 1u) ? v[i] : 0.0f;
                }
                // branch-free: a row past M (or the ablation build's "no output stores") gets an out-of-range offset and stores nothing
                const bool st_live = row < a.M && ((DETR_ABLATE & 64) == 0 || v[0] == 12345.678f);
                const u32x4 ov = {f32_to_bf16_pair(v[0], v[1]), f32_to_bf16_pair(v[2], v[3]), f32_to_bf16_pair(v[4], v[5]), f32_to_bf16_pair(v[6], v[7])};
                dstC.st16(st_live ? (unsigned)row * ldc2 + (unsigned)((n0 + sw * 64 + ecg * 8) * 2) : BUF_OOB, ov);
                if constexpr (MBO && !DETR_STREAM_PROF)
                    dstB.st1(st_live ? (unsigned)row * ldb1 + (unsigned)(((n0 + sw * 64) >> 3) + ecg) : BUF_OOB, (unsigned char)bf16x8_gt0_bits(ov[0], ov[1], ov[2], ov[3]));
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            STREAM_TICK(4);                         // epilogue items: LDS reads, residual / mask waits, stores issued
        }
    };

    int rt = p * WPS + wr;
    uint4 a0[KK], a1[KK];
    load_a(rt, 0, a0);                             // first A rows: in flight while the B slice is staged

    // ---- prologue: B slice -> LDS as [n][k] (the only workgroup barrier of the kernel) ---------------------------
    const int g0 = nt * SL * NW * 64;              // first column of the group
    if (BKC) {
        for (int c = tid; c < SL * NW * 64 * (K / 8); c += THREADS) {
            const int n = c / (K / 8), kc = c - n * (K / 8);
            const uint4 v = *reinterpret_cast<const uint4 *>(a.B + (long long)(g0 + n) * a.ldb + kc * 8);
            *reinterpret_cast<uint4 *>(&sm.B[n >> 6][n & 63][kc * 8]) = v;
        }
    } else {
        // [k][n] weights keep their orientation: 16-byte chunk (k, 8 columns) -> sub-block (k / 4, n / 16) of the transpose-read image,
        // ONE 16-byte LDS store (round 3 scattered it into [n][k] with eight 2-byte stores: ~8 us of a 39 us K = 256 launch)
        for (int c = tid; c < K * 8 * SL * NW; c += THREADS) {
            const int k = c / (8 * SL * NW), nc = c - k * (8 * SL * NW);
            const uint4 v = *reinterpret_cast<const uint4 *>(a.B + (long long)k * a.ldb + g0 + nc * 8);
            const int ncl = nc & 7;
            unsigned short *img = &sm.B[nc >> 3][0][0];
            *reinterpret_cast<uint4 *>(img + (((k >> 2) * 4 + (ncl >> 1)) * 64 + (k & 3) * 16 + (ncl & 1) * 8)) = v;
        }
    }
    __syncthreads();

#if DETR_STREAM_PROF
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
    prof_last = prof_t0;
#endif
    while (rt < a.row_tiles) {
        strip(rt, a0, a1);
        rt += stride;
        if (NC == 1) {
            if (rt >= a.row_tiles) break;
            strip(rt, a1, a0);
            rt += stride;
        }
    }
#if DETR_STREAM_PROF
    if (lane == 0 && a.mbits_out) {                // [workgroup][wave][8]: sections 0-4, 5 = prologue end, 6 = loop total, 7 = between strips
        unsigned long long *o = reinterpret_cast<unsigned long long *>(a.mbits_out) + ((size_t)blockIdx.x * WAVES + wave) * 8;
        prof_acc[6] = __builtin_readcyclecounter() - prof_t0;
        prof_acc[5] = prof_t0;
        for (int k = 0; k < 8; ++k) o[k] = prof_acc[k];
    }
#endif
}

// Host side: eligibility is decided by the caller (gemm_f32.hip); here the slice grouping and the grid.
template <int K, int SL, bool EXT = false, int NW = 1>
static void launch_gemm_stream_sl(StreamArgs a, bool bkc, hipStream_t s, bool mask_bits = false) {
    constexpr int WAVES = StreamSmem<K, SL, NW>::WAVES;
    a.n_tiles = a.N / (64 * SL * NW);
    a.row_tiles = (a.M + 31) / 32;
    const int wgs_per_cu = StreamOcc<K, SL, NW>::VALUE;
    int q = (256 * wgs_per_cu) / (8 * a.n_tiles);
    const int qmax = a.row_tiles / (8 * (WAVES / SL) * 2);      // at least two strips per wave
    if (q > qmax) q = qmax;
    if (q < 1) q = 1;
    a.q = q;
    const dim3 grid((unsigned)(8 * a.n_tiles * q));
    const bool r = a.res != nullptr, m = a.mask != nullptr;
#define DETR_STREAM_LAUNCH(BK_, R_, M_) hipLaunchKernelGGL((gemm_stream_bf16_kernel<K, BK_, R_, M_, SL, EXT, NW>), grid, dim3(64 * WAVES), 0, s, a)
    if (a.mbits_out) {          // mask bits out: the forward form only -- gemm_stream_eligible() sends nothing else here ([k][n] weights + residual,
                                // no mask in, no extended epilogue; any other request with bits out runs on the tile engine)
        if constexpr (!EXT) hipLaunchKernelGGL((gemm_stream_bf16_kernel<K, false, true, 0, SL, false, NW, true>), grid, dim3(64 * WAVES), 0, s, a);
    } else if (m && mask_bits) {       // bit-packed mask (the input gradients of the bottleneck blocks' first 1x1 convolution: BKC layout)
        if (bkc) { if (r) DETR_STREAM_LAUNCH(true, true, 2); else DETR_STREAM_LAUNCH(true, false, 2); }
        else { if (r) DETR_STREAM_LAUNCH(false, true, 2); else DETR_STREAM_LAUNCH(false, false, 2); }
    } else if (bkc) {
        if (r && m) DETR_STREAM_LAUNCH(true, true, 1);
        else if (r) DETR_STREAM_LAUNCH(true, true, 0);
        else if (m) DETR_STREAM_LAUNCH(true, false, 1);
        else DETR_STREAM_LAUNCH(true, false, 0);
    } else {
        if (r && m) DETR_STREAM_LAUNCH(false, true, 1);
        else if (r) DETR_STREAM_LAUNCH(false, true, 0);
        else if (m) DETR_STREAM_LAUNCH(false, false, 1);
        else DETR_STREAM_LAUNCH(false, false, 0);
    }
#undef DETR_STREAM_LAUNCH
}

template <int K>
static void launch_gemm_stream(StreamArgs a, bool bkc, hipStream_t s, bool mask_bits = false) {
    // slices per workgroup, by measurement (scripts/experiments/ablate_stream.py with DETR_HIP_STREAM_SL = 1 / 2 / 4, the tuning
    // hook below; SL 1 -> chosen): M534400 N256 K64 130 -> 123 us (+mask 180 -> 171), M133600 N512 K128 +res 95 -> 87 (SL 2),
    // +res +mask 116 -> 95 (SL 4), M133600 N512 K256 144 -> 126 (SL 2); M33600 N1024 K256 stays at SL 1 (52 vs 57 us)
    // (in the step, HIP events: K = 256 without a residual / mask epilogue is SLOWER with 2 slices -- 102 KB of LDS, one workgroup per
    //  CU: M133600 N512 0.091 -> 0.120 ms -- so K = 256 groups only the epilogue-heavy form)
    // DETR_HIP_STREAM_NW=1: one column slice per wave everywhere; otherwise two wherever that form exists (K = 256, N % 128 == 0):
    // M33600 N1024 +res 63.8 -> 59.5 us, +res +mask 71.0 -> 64.3, M133600 N512 +res +mask 126 -> 107, M8400 N2048 28.3 -> 26.9, M534400 N128 122 -> 113
    const int nw_mode = tune(T_STREAM_NW);
    const bool nw2 = K == 256 && a.N % 128 == 0 && nw_mode != 1;
    if (a.alpha != 1.0f || a.drop_scale != 0.0f) {      // the extended epilogue exists for K = 256, one slice per workgroup
        if constexpr (K == 256) {
            if (nw2) launch_gemm_stream_sl<256, 1, true, 2>(a, bkc, s, mask_bits);
            else launch_gemm_stream_sl<256, 1, true>(a, bkc, s, mask_bits);
        }
        return;
    }
    if constexpr (K == 256) {
        if (nw2 && tune(T_STREAM_SL) == 0) { launch_gemm_stream_sl<256, 1, false, 2>(a, bkc, s, mask_bits); return; }
    }
    if constexpr (K == 128) {       // two slice pairs per 8-wave workgroup: M133600 N512 +res 80.8 -> 66.5 us, +res +mask 89.4 -> 85.9
        // (one pair per 4-wave workgroup: 70.6 / 88.9; K = 64 loses with either form: 112 -> 117 us, it is bound by its output stores)
        if (nw_mode != 1 && a.N % 256 == 0 && tune(T_STREAM_SL) == 0) { launch_gemm_stream_sl<128, 2, false, 2>(a, bkc, s, mask_bits); return; }
    }
    int sl = (K == 64) ? 4 : (K == 128 ? ((a.res && a.mask) ? 4 : 2) : ((a.res && a.mask && a.N <= 512) ? 2 : 1));
    const int force = tune(T_STREAM_SL);
    if (force == 1 || force == 2 || force == 4) sl = force;
    while (sl > 1 && (a.N % (64 * sl) != 0 || (int)sizeof(StreamSmem<K, 1>) + (sl - 1) * 64 * (K + 8) * 2 > 160 * 1024)) sl >>= 1;
    if (sl == 4) {
        if constexpr (K <= 128) { launch_gemm_stream_sl<K, 4>(a, bkc, s, mask_bits); return; }
        sl = 2;
    }
    if (sl == 2) launch_gemm_stream_sl<K, 2>(a, bkc, s, mask_bits);
    else launch_gemm_stream_sl<K, 1>(a, bkc, s, mask_bits);
}

}  // namespace detr
