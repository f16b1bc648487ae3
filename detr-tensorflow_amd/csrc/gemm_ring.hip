// gemm_ring.hip -- host side of the 8-wave LDS-DMA ring GEMM (gemm_ring.h): tile plan (row-balanced grid), instantiations, launch.
// Reference lines it serves: the 1x1 convolutions of resnet_backbone.py:119-135 (forward and input gradient) and the FFN
// Linear layers of transformer.py:172-177 at their training shapes (M = B*H*W >= 4096).
#include "gemm_ring.h"

namespace detr {

// Cost model of one candidate grid (cycles of the slowest CU, up to a common factor).  A workgroup's K loop is bounded by its
// MFMA issue (2 waves per SIMD, 32 cycles per 32x32x16 MFMA) and by the bytes it pulls through the CU's vector-memory path
// (RING_BPC bytes per clock: measured order of magnitude for LDS-DMA out of L2, profiles/r05_*), plus a fixed prologue /
// epilogue term; workgroups beyond 256 run in further rounds.
static constexpr double RING_BPC = 24.0;
static constexpr double RING_FIXED = 6000.0;

static double ring_cost(int M, int N, int K, int bn, int tile_rows, int &tm, int &wgs) {
    const int tiles_m = cdiv(M, tile_rows), tiles_n = cdiv(N, bn);
    const int blocks = cdiv(tile_rows, 32);
    tm = cdiv(blocks, 2);
    const int tn = bn / 128;
    wgs = tiles_m * tiles_n;
    const double mfma = (double)tm * tn * (K / 16) * 32.0 * 2.0;
    const double traffic = (double)(tile_rows + bn) * K * 2.0 / RING_BPC;
    const double epi = (double)tile_rows * bn * 4.0 / RING_BPC;
    const double per = (mfma > traffic ? mfma : traffic) + epi + RING_FIXED;
    return per * cdiv(wgs, 256);
}

bool gemm_ring_plan(int M, int N, int K, RingPlan &p) {
    if (M < 1 || N < 128 || K < RING_BK || K % RING_BK != 0) return false;
    const int force_bn = tune(T_RING_BN), force_wgs = tune(T_RING_WGS), force_ns = tune(T_RING_NS), force_rows = tune(T_RING_ROWS);
    double best = 1e300;
    bool found = false;
    for (int bn = 128; bn <= 256; bn += 128) {
        if (force_bn && bn != force_bn) continue;
        if (bn == 256 && N < 256) continue;
        if (bn == 128 && N > 256 && N % 256 == 0 && !force_bn) continue;     // wide outputs: 256-column panels (half the A re-reads)
        const int tiles_n = cdiv(N, bn);
        for (int target = 64; target <= 256 * 24; target += (target < 256 ? 32 : 64)) {
            if (force_wgs && target != force_wgs) continue;
            int tiles_m = target / tiles_n;
            if (tiles_m < 1) continue;
            int rows = cdiv(M, tiles_m);
            rows = (rows + 3) & ~3;
            if (rows < 8) rows = 8;
            if (force_rows >= 8) rows = (force_rows + 3) & ~3;               // (sweeps: scripts/experiments/ring_sweep.py)
            if (rows > 256) continue;
            int tm, wgs;
            const double c = ring_cost(M, N, K, bn, rows, tm, wgs);
            if (c < best) {
                best = c;
                found = true;
                p.tm = tm; p.tn = bn / 128; p.tile_rows = rows; p.tiles_m = cdiv(M, rows); p.tiles_n = tiles_n; p.wgs = wgs;
            }
        }
    }
    if (!found) return false;
    const int bn = 128 * p.tn;
    p.a_rows8 = (p.tile_rows + 7) & ~7;
    p.stage_bytes = (p.a_rows8 + bn) * RING_STAGE_ROW;
    const int epi_bytes = 8 * 32 * (bn / 4 + 4) * 4;                       // the epilogue's wave-private staging strips (gemm_core.h StageCfg)
    p.ns = (3 * p.stage_bytes + 1024 <= 160 * 1024) ? 3 : 2;
    if (force_ns == 2 || (force_ns == 3 && p.ns >= 3)) p.ns = force_ns;
    p.dump_off = p.ns * p.stage_bytes;
    const int ring_bytes = p.dump_off + 1024;
    p.lds_bytes = ring_bytes > epi_bytes ? ring_bytes : epi_bytes;
    p.cost = best;
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
    hipLaunchKernelGGL((gemm_ring_kernel<TM, TN, BKC, NS>), dim3((unsigned)wgs), dim3(RING_THREADS), (size_t)lds, s, ra);
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
