// gemm_ring.hip -- host side of the 8-wave LDS-DMA ring GEMM (gemm_ring.h): tile plan (row-balanced grid), instantiations, launch.
// Reference lines it serves: the 1x1 convolutions of resnet_backbone.py:119-135 (forward and input gradient) and the FFN
// Linear layers of transformer.py:172-177 at their training shapes (M = B*H*W >= 4096).
#include "gemm_ring.h"

namespace detr {

// Tile plan.  Measured (scripts/experiments/ring_sweep.py, profiles/r05_ring_sweep.txt: every eligible shape of the step x row pitch x
// column panel x ring depth): one rule wins on all of them within 2-5 % of the per-shape optimum --
//   * 128-column panels (TN = 1) and a row pitch of 176 (TM = 3: 96 + 80 rows for the two row waves) with a TWO-stage ring: the
//     workgroup needs 77 KB of LDS, so TWO workgroups share a CU and one's epilogue (the output / residual / mask streams are as
//     many bytes as the A operand at K = 512) runs under the other's K loop.  256-column panels with a 3-stage ring (one
//     workgroup per CU, A read once) lose 5-30 %: their epilogue overlaps nothing;
//   * when the problem does not fill the 512 slots even once (M = 8400 with K = 2048: 212 workgroups), one round of ~200
//     workgroups with a THREE-stage ring (one workgroup per CU anyway: the deeper ring hides more of the L2 latency).
// DETR_HIP_RING_BN / _NS / _ROWS / _WGS force the pieces of the plan (sweeps, tests).
bool gemm_ring_plan(int M, int N, int K, RingPlan &p) {
    if (M < 1 || N < 128 || K < RING_BK || K % RING_BK != 0) return false;
    const int force_bn = tune(T_RING_BN), force_wgs = tune(T_RING_WGS), force_ns = tune(T_RING_NS), force_rows = tune(T_RING_ROWS);
    int bn = 128;
    if (force_bn == 256 && N >= 256) bn = 256;
    const int tiles_n = cdiv(N, bn);
    int rows = 176, ns = 2;
    if (cdiv(M, 176) * tiles_n < 320) {                  // a single round: ~200 workgroups, three stages
        rows = (int)(((long long)M * tiles_n + 199) / 200);
        rows = (rows + 3) & ~3;
        if (rows > 176) rows = 176;
        ns = 3;
    }
    if (force_wgs > 0) {
        const int tm_ = force_wgs / tiles_n > 0 ? force_wgs / tiles_n : 1;
        rows = (cdiv(M, tm_) + 3) & ~3;
    }
    if (force_rows >= 8) rows = (force_rows + 3) & ~3;
    if (rows < 8) rows = 8;
    if (rows > 256) return false;
    p.tile_rows = rows;
    p.tm = cdiv(cdiv(rows, 32), 2);
    p.tn = bn / 128;
    p.tiles_m = cdiv(M, rows);
    p.tiles_n = tiles_n;
    p.wgs = p.tiles_m * tiles_n;
    p.a_rows8 = (rows + 7) & ~7;
    p.stage_bytes = (p.a_rows8 + bn) * RING_STAGE_ROW;
    const int epi_bytes = 8 * 32 * (bn / 4 + 4) * 4;                       // the epilogue's wave-private staging strips (gemm_core.h StageCfg)
    if (3 * p.stage_bytes + 1024 > 160 * 1024) ns = 2;
    if (force_ns == 2 || (force_ns == 3 && 3 * p.stage_bytes + 1024 <= 160 * 1024)) ns = force_ns;
    p.ns = ns;
    p.dump_off = p.ns * p.stage_bytes;
    const int ring_bytes = p.dump_off + (p.a_rows8 < 64 * p.tm ? 1024 : 0);      // (no piece lies past a full-capacity A image: no dump area)
    p.lds_bytes = ring_bytes > epi_bytes ? ring_bytes : epi_bytes;
    p.cost = 0.0;
    return p.lds_bytes <= 160 * 1024;
}

template <int TM, int TN, bool BKC, int NS>
static int ring_launch_one(const RingArgs &ra, int wgs, int lds, hipStream_t s) {
    static int reserved = 0;                            // largest dynamic-LDS size this instantiation has been granted
    if (lds > reserved) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_ring_kernel<TM, TN, BKC, NS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        DETR_REQUIRE(e == hipSuccess, "gemm (ring): cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e));
        reserved = 160 * 1024;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<TM, TN, BKC, NS>), dim3((unsigned)wgs), dim3(RING_THREADS), (size_t)lds + ((DETR_ABLATE & 64) != 0 ? 1024 : 0), s, ra);
    return 0;
}

template <int TM, int TN>
static int ring_launch_tt(const RingArgs &ra, const RingPlan &p, bool bk, hipStream_t s) {
    if (p.ns == 3) return bk ? ring_launch_one<TM, TN, true, 3>(ra, p.wgs, p.lds_bytes, s) : ring_launch_one<TM, TN, false, 3>(ra, p.wgs, p.lds_bytes, s);
    return bk ? ring_launch_one<TM, TN, true, 2>(ra, p.wgs, p.lds_bytes, s) : ring_launch_one<TM, TN, false, 2>(ra, p.wgs, p.lds_bytes, s);
}

int gemm_ring_launch(const GemmArgs &g, bool bk, const RingPlan &p, hipStream_t s) {
    RingArgs ra;
    ra.g = g;
    ra.g.tiles_m = p.tiles_m;
    ra.g.tiles_n = p.tiles_n;
    ra.tile_rows = p.tile_rows;
    ra.a_rows8 = p.a_rows8;
    ra.stage_bytes = p.stage_bytes;
    ra.dump_off = p.dump_off;
    ra.trace_off = p.lds_bytes;
    ra.ablate = tune(T_RING_ABLATE);
    int rc = -1;
    switch (p.tm * 2 + (p.tn - 1)) {
        case 2: rc = ring_launch_tt<1, 1>(ra, p, bk, s); break;
        case 3: rc = ring_launch_tt<1, 2>(ra, p, bk, s); break;
        case 4: rc = ring_launch_tt<2, 1>(ra, p, bk, s); break;
        case 5: rc = ring_launch_tt<2, 2>(ra, p, bk, s); break;
        case 6: rc = ring_launch_tt<3, 1>(ra, p, bk, s); break;
        case 7: rc = ring_launch_tt<3, 2>(ra, p, bk, s); break;
        case 8: rc = ring_launch_tt<4, 1>(ra, p, bk, s); break;
        case 9: rc = ring_launch_tt<4, 2>(ra, p, bk, s); break;
        default: DETR_REQUIRE(false, "gemm (ring): no instantiation for TM=%d TN=%d", p.tm, p.tn);
    }
    return rc;
}

// fp32 plan: the f32 MFMA runs at 1/16 of the bf16 rate, so these GEMMs are MFMA-bound and what counts is (a) whole 32-row blocks
// (a SIMD's time is the number of blocks its two row waves issue) and (b) the same number of blocks on every CU: the pitch is the
// multiple of 32 in 64 .. 192 that minimises ceil(workgroups / 256) x blocks per workgroup (ties: the larger pitch -- fewer re-reads of
// the weights); 128-column panels, two stages: <= 80 KB of LDS, two workgroups per CU.
bool gemm_ring_f32_plan(int M, int N, int K, RingPlan &p) {
    if (M < 1 || N < 128 || K < 32 || K % 32 != 0) return false;
    const int bn = 128, tiles_n = cdiv(N, bn);
    const int force_rows = tune(T_RING_ROWS);
    long long best = -1;
    int rows = 64;
    for (int r = 64; r <= 192; r += 32) {
        const long long c = (long long)cdiv((long long)cdiv(M, r) * tiles_n, 256) * (r / 32);
        if (best < 0 || c <= best) { best = c; rows = r; }
    }
    if (force_rows >= 8) rows = (force_rows + 3) & ~3;
    if (rows > 192) return false;
    p.tile_rows = rows;
    p.tm = cdiv(cdiv(rows, 32), 2);
    p.tn = 1;
    p.tiles_m = cdiv(M, rows);
    p.tiles_n = tiles_n;
    p.wgs = p.tiles_m * tiles_n;
    p.a_rows8 = (rows + 7) & ~7;
    p.stage_bytes = (p.a_rows8 + bn) * RING_STAGE_ROW;
    p.ns = 2;
    p.dump_off = p.ns * p.stage_bytes;
    const int epi_bytes = 8 * 32 * (bn / 4 + 4) * 4;
    const int ring_bytes = p.dump_off + (p.a_rows8 < 64 * p.tm ? 1024 : 0);       // (no piece lies past a full-capacity image: no dump area)
    p.lds_bytes = ring_bytes > epi_bytes ? ring_bytes : epi_bytes;
    p.cost = (double)best;
    return p.lds_bytes <= 160 * 1024;
}

template <int TM, bool BKC>
static int ring_f32_launch_one(const RingArgs &ra, int wgs, int lds, hipStream_t s) {
    static bool reserved = false;
    if (!reserved) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_ring_f32_kernel<TM, 1, BKC, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        DETR_REQUIRE(e == hipSuccess, "gemm (ring, fp32): cannot reserve %d bytes of LDS: %s", lds, hipGetErrorString(e));
        reserved = true;
    }
    hipLaunchKernelGGL((gemm_ring_f32_kernel<TM, 1, BKC, 2>), dim3((unsigned)wgs), dim3(RING_THREADS), (size_t)lds, s, ra);
    return 0;
}

int gemm_ring_f32_launch(const GemmArgs &g, bool bk, const RingPlan &p, hipStream_t s) {
    RingArgs ra;
    ra.g = g;
    ra.g.tiles_m = p.tiles_m;
    ra.g.tiles_n = p.tiles_n;
    ra.tile_rows = p.tile_rows;
    ra.a_rows8 = p.a_rows8;
    ra.stage_bytes = p.stage_bytes;
    ra.dump_off = p.dump_off;
    ra.trace_off = 0;
    ra.ablate = tune(T_RING_ABLATE);
    DETR_REQUIRE(p.tn == 1 && p.ns == 2 && p.tm >= 1 && p.tm <= 3, "gemm (ring, fp32): no instantiation for TM=%d TN=%d NS=%d", p.tm, p.tn, p.ns);
    if (p.tm == 1) return bk ? ring_f32_launch_one<1, true>(ra, p.wgs, p.lds_bytes, s) : ring_f32_launch_one<1, false>(ra, p.wgs, p.lds_bytes, s);
    if (p.tm == 2) return bk ? ring_f32_launch_one<2, true>(ra, p.wgs, p.lds_bytes, s) : ring_f32_launch_one<2, false>(ra, p.wgs, p.lds_bytes, s);
    return bk ? ring_f32_launch_one<3, true>(ra, p.wgs, p.lds_bytes, s) : ring_f32_launch_one<3, false>(ra, p.wgs, p.lds_bytes, s);
}

template <int TM, int TN, int NS>
static int ring_wgrad_launch_one(const RingArgs &ra, int wgs, hipStream_t s) {
    constexpr int LDS = NS * (64 * TM + 128 * TN) * RING_STAGE_ROW;
    static_assert(LDS <= 160 * 1024, "ring does not fit the LDS");
    static bool reserved = false;
    if (!reserved) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_ring_wgrad_kernel<TM, TN, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        DETR_REQUIRE(e == hipSuccess, "gemm (ring wgrad): cannot reserve LDS: %s", hipGetErrorString(e));
        reserved = true;
    }
    hipLaunchKernelGGL((gemm_ring_wgrad_kernel<TM, TN, NS>), dim3((unsigned)wgs), dim3(RING_THREADS), (size_t)LDS, s, ra);
    return 0;
}

// bm x bn: 128 x 128 (the product), 128 x 256 / 256 x 128 (DETR_HIP_RING_WTILE experiments)
int gemm_ring_wgrad_launch(const GemmArgs &g, int bm, int bn, int split, hipStream_t s) {
    RingArgs ra;
    ra.g = g;
    ra.g.tiles_m = cdiv(g.M, bm);
    ra.g.tiles_n = cdiv(g.N, bn);
    ra.tile_rows = ra.a_rows8 = ra.stage_bytes = ra.dump_off = ra.trace_off = 0;
    ra.ablate = tune(T_RING_ABLATE);
    const int wgs = ra.g.tiles_m * ra.g.tiles_n * split;
    if (bm == 128 && bn == 128) return tune(T_RING_NS) == 3 ? ring_wgrad_launch_one<2, 1, 3>(ra, wgs, s) : ring_wgrad_launch_one<2, 1, 4>(ra, wgs, s);
    if (bm == 128 && bn == 256) return ring_wgrad_launch_one<2, 2, 3>(ra, wgs, s);
    if (bm == 256 && bn == 128) return ring_wgrad_launch_one<4, 1, 3>(ra, wgs, s);
    DETR_REQUIRE(false, "gemm (ring wgrad): no instantiation for %d x %d tiles", bm, bn);
    return -1;
}

}  // namespace detr

// Plan query (tests, tuning scripts): fills out[0..7] = {tm, tn, ns, tile_rows, tiles_m, tiles_n, workgroups, lds bytes}; returns 1 when
// the shape has a plan, 0 when not.  Pure host arithmetic.
extern "C" int detr_hip_gemm_ring_plan(int32_t M, int32_t N, int32_t K, int32_t *out) {
    detr::RingPlan p;
    if (!detr::gemm_ring_plan(M, N, K, p)) return 0;
    if (out) {
        out[0] = p.tm; out[1] = p.tn; out[2] = p.ns; out[3] = p.tile_rows; out[4] = p.tiles_m; out[5] = p.tiles_n; out[6] = p.wgs; out[7] = p.lds_bytes;
    }
    return 1;
}

#if (DETR_ABLATE & 64) != 0
namespace detr { __device__ long long ring_trace[3][RING_TR_N]; }
extern "C" int detr_hip_debug_ring_trace(long long *out) {       // experiment builds only: 3 x RING_TR_N stamps of the last ring GEMM launch
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(detr::ring_trace), sizeof(detr::ring_trace)) == hipSuccess ? 0 : -1;
}
#endif
