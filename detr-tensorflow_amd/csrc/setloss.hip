// setloss.hip -- the Hungarian set loss on the device (no host round trip):
//   K12 match_cost   : [Q, n] cost matrix per (level, image)      hungarian_matching.py:171-195
//   K13 assign       : exact rectangular assignment, one wave64 per problem, cost tile in LDS
//                      (replaces tf.numpy_function -> scipy.optimize.linear_sum_assignment,
//                       hungarian_matching.py:27-46,197); shortest-augmenting-path / JV with
//                       double-precision duals exactly as SciPy's solver, lanes over the
//                       prediction columns, wave-butterfly argmin
//   K14 set_loss_*   : weighted CE, L1, GIoU (clipped xyxy), metrics, and the gradients w.r.t.
//                      logits / boxes                              loss.py:37-96, bbox.py:29-124,171-183
// Compiled with -ffp-contract=off so that the box arithmetic rounds like the op-by-op reference.
#include "common.h"
#include <limits.h>
#include <type_traits>

namespace detr {

struct SetLossArgs {
    int levels, B, Q, C, R;
    const float *logits; long long sL_l, sL_b, sL_q;
    const float *boxes;  long long sB_l, sB_b, sB_q;
    const float *t_bbox; const long long *t_class;
    int background_class;
};

__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

struct XYXY { float x1, y1, x2, y2; };

__device__ __forceinline__ XYXY to_xyxy(float cx, float cy, float w, float h) {
    // bbox.py:171-183 (clip to [0,1])
    XYXY r;
    r.x1 = clip01(cx - w / 2.0f);
    r.y1 = clip01(cy - h / 2.0f);
    r.x2 = clip01(cx + w / 2.0f);
    r.y2 = clip01(cy + h / 2.0f);
    return r;
}

__device__ __forceinline__ float giou_of(const XYXY &p, const XYXY &t) {
    // bbox.py:29-105 + hungarian_matching.py:186-192 / loss.py:84-91
    const float iw = fmaxf(fminf(p.x2, t.x2) - fmaxf(p.x1, t.x1), 0.0f);
    const float ih = fmaxf(fminf(p.y2, t.y2) - fmaxf(p.y1, t.y1), 0.0f);
    const float inter = iw * ih;
    const float area_a = (p.x2 - p.x1) * (p.y2 - p.y1);
    const float area_b = (t.x2 - t.x1) * (t.y2 - t.y1);
    const float uni = area_a + area_b - inter;
    const float iou = inter / uni;
    const float cw = fmaxf(fmaxf(p.x2, t.x2) - fminf(p.x1, t.x1), 0.0f);
    const float ch = fmaxf(fmaxf(p.y2, t.y2) - fminf(p.y1, t.y1), 0.0f);
    const float area = cw * ch;
    return iou - (area - uni) / area;
}

__device__ __forceinline__ int header_n(const float *t_bbox, int b, int R) {
    int n = (int)t_bbox[(long long)b * R * 4];
    if (n < 0) n = 0;
    if (n > R - 1) n = R - 1;
    return n;
}

// ------------------------------------------------------------------------------------------------
// K12: cost matrix.  One workgroup per (level, image).
// ------------------------------------------------------------------------------------------------
constexpr int SL_MAXQ = 512;
constexpr int SL_MAXR = 128;

constexpr int SL_QCHUNK = 8;      // queries per workgroup of the cost / sums / gradient kernels (two per wave)

__global__ __launch_bounds__(256) void match_cost_kernel(SetLossArgs a, float *__restrict__ cost) {
    __shared__ float s_max[SL_QCHUNK], s_sum[SL_QCHUNK];
    __shared__ float s_t[SL_MAXR][4];
    __shared__ int s_cls[SL_MAXR];
    const int p = blockIdx.x;
    const int lv = p / a.B, b = p % a.B;
    const int n = header_n(a.t_bbox, b, a.R);
    const int ldc = a.R - 1;
    if (n == 0) return;
    const float *lg = a.logits + lv * a.sL_l + b * a.sL_b;
    const float *bx = a.boxes + lv * a.sB_l + b * a.sB_b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const float *t = a.t_bbox + ((long long)b * a.R + 1 + j) * 4;
        s_t[j][0] = t[0]; s_t[j][1] = t[1]; s_t[j][2] = t[2]; s_t[j][3] = t[3];
        s_cls[j] = (int)a.t_class[(long long)b * a.R + 1 + j];
    }
    // blockIdx.y: a chunk of SL_QCHUNK queries of the problem (one workgroup per problem walked its 100 queries in 25 rounds of
    // wave reductions: 45-80 us of pure latency on the path between the forward and the backward pass)
    const int q0 = blockIdx.y * SL_QCHUNK, q1 = min(a.Q, q0 + SL_QCHUNK);
    for (int q = q0 + wave; q < q1; q += 4) {
        const float *row = lg + q * a.sL_q;
        float mx = -INFINITY;
        for (int c = lane; c < a.C; c += 64) mx = fmaxf(mx, row[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int c = lane; c < a.C; c += 64) s += expf(row[c] - mx);
        s = wave_sum(s);
        if (lane == 0) { s_max[q - q0] = mx; s_sum[q - q0] = s; }
    }
    __syncthreads();
    float *out = cost + (long long)p * a.Q * ldc;
    for (int idx = threadIdx.x; idx < (q1 - q0) * n; idx += blockDim.x) {
        const int ql = idx / n, j = idx - ql * n;
        const int q = q0 + ql;
        const float *pb = bx + q * a.sB_q;
        const float pcx = pb[0], pcy = pb[1], pw = pb[2], ph = pb[3];
        const float tcx = s_t[j][0], tcy = s_t[j][1], tw = s_t[j][2], th = s_t[j][3];
        const int cls = s_cls[j];
        float prob = 0.0f;
        if (cls >= 0 && cls < a.C) prob = expf(lg[q * a.sL_q + cls] - s_max[ql]) / s_sum[ql];
        const float cost_class = -prob;
        const float cost_bbox = fabsf(pcx - tcx) + fabsf(pcy - tcy) + fabsf(pw - tw) + fabsf(ph - th);
        const float cost_giou = -giou_of(to_xyxy(pcx, pcy, pw, ph), to_xyxy(tcx, tcy, tw, th));
        out[q * ldc + j] = 5.0f * cost_bbox + 1.0f * cost_class + 2.0f * cost_giou;
    }
}

// ------------------------------------------------------------------------------------------------
// K13: rectangular LSAP, one wave per problem.  Rows of the (transposed) problem = targets,
// columns = predictions; column j lives on lane j&63, slot j>>6.
// ------------------------------------------------------------------------------------------------
// min over the 64 lanes of a double, returned wave-uniform.  Row-shift DPP scan inside each row of
// 16 lanes, then row_bcast:15 / row_bcast:31 fold the four rows into lane 63 (the gfx9 wave64
// reduction idiom); ~20 VALU ops instead of 12 dependent ds_bpermute round trips.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_min_step(double x, int &tlo, int &thi) {
    // lanes this stage does not write (no DPP source, or outside ROW_MASK) keep what an earlier stage left in (thi, tlo):
    // an older partial minimum of genuine elements, harmless for a minimum -- so no re-initialisation with +inf per stage
    tlo = __builtin_amdgcn_update_dpp(tlo, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    thi = __builtin_amdgcn_update_dpp(thi, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    // raw v_min_f64: the operands are never NaN here (the cost matrix is screened, +inf - finite stays +inf), so
    // the two v_max_f64 canonicalisations fmin() would add to every step of the dependent chain are dead weight
    const double y = __hiloint2double(thi, tlo);
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ double wave_min_f64_dpp(double x) {
    int tlo = __double2loint(x), thi = __double2hiint(x);
    x = dpp_min_step<0x111, 0xf>(x, tlo, thi);   // row_shr:1
    x = dpp_min_step<0x112, 0xf>(x, tlo, thi);   // row_shr:2
    x = dpp_min_step<0x114, 0xf>(x, tlo, thi);   // row_shr:4
    x = dpp_min_step<0x118, 0xf>(x, tlo, thi);   // row_shr:8
    x = dpp_min_step<0x142, 0xa>(x, tlo, thi);   // row_bcast:15 into rows 1 and 3
    x = dpp_min_step<0x143, 0xc>(x, tlo, thi);   // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}

// CD: the transposed cost tile is held in LDS as doubles (Q <= 128: 99 x 129 x 8 B = 102 KB), which takes the
// float -> double conversion out of every row scan.
//
// Shape of the search (LAPJV's SCAN / TODO form of the Dijkstra, same labels as SciPy's loop): a wave-wide minimum is
// only taken when the set of live columns whose label EQUALS the current minimum (`todo`) is empty.  Measured with
// s_memtime (DETR_ASSIGN_PROF builds, scripts/experiments/assign_probe.py) on the 99-target problems of the bench step:
// the loop is bound by the latency of ONE wave's dependent instructions, ~880 cycles per scanned row when every row scan
// is followed by a minimum + owner selection (6 DPP stages of 64-bit moves and v_min_f64, then ballots, v_readlane and the
// SALU <-> VALU round trips between them), of which the row scan itself is ~100.  About half of all scanned columns are
// exact ties with the current minimum (tight edges: reduced cost 0), so they are scanned straight from `todo` without
// a reduction.  The column sets (live, todo, free) are wave-uniform 64-bit masks (one SGPR pair per slot), used directly
// as lane conditions and picked from with s_ff1.
// next column to scan: the lowest of the lowest non-empty slot of `todo` (any order is a valid Dijkstra order: all of them
// carry the final label minVal).  One branch per slot, constant slot index inside it (no selects between SGPR pairs).
template <int S, int N>
struct AssignPick {
    static __device__ __forceinline__ void run(unsigned long long (&todo)[N], unsigned long long (&live)[N],
                                               const unsigned long long (&freecol)[N], const int (&row4col)[N], int &jstar,
                                               int &r4, bool &isfree) {
        if (S + 1 == N || todo[S]) {
            const int l = ((int)__ffsll((long long)todo[S]) - 1) & 63;
            const unsigned long long bit = 1ull << l;
            todo[S] &= ~bit;
            live[S] &= ~bit;
            isfree = (freecol[S] & bit) != 0ull;
            r4 = __builtin_amdgcn_readlane(row4col[S], l);
            jstar = 64 * S + l;
        } else {
            AssignPick<(S + 1 < N ? S + 1 : S), N>::run(todo, live, freecol, row4col, jstar, r4, isfree);
        }
    }
};

#ifndef DETR_ASSIGN_PROF
#define DETR_ASSIGN_PROF 0       // 1: s_memtime segment counters of the step loop, written over pred_for_tgt[0..15] (probe builds only)
#endif
#if DETR_ASSIGN_PROF
#define ASSIGN_TICK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[k] += t_ - prof_last; prof_last = t_; } while (0)
#else
#define ASSIGN_TICK(k) do { } while (0)
#endif
template <int MAXCPL, bool CD>
__global__ __launch_bounds__(64) void assign_kernel(const float *__restrict__ cost, int Q, int ldc,
                                                    const float *__restrict__ t_bbox, int B, int R,
                                                    int *__restrict__ tgt_for_pred, int *__restrict__ pred_for_tgt,
                                                    int *__restrict__ status) {
    using CT = typename std::conditional<CD, double, float>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int p = blockIdx.x, b = p % B, lane = threadIdx.x;
    const int nmax = (R + 1) & ~1;
    const int n = header_n(t_bbox, b, R);
    const int cpl = (Q + 63) >> 6;
    const int Qs = CD ? MAXCPL * 64 + 1 : cpl * 64 + 1;   // row stride of the transposed cost (odd: the column-wise staging
                                        // writes spread over the banks); slot s of lane l sits at 64*s + l
    double *u0 = reinterpret_cast<double *>(smem);
    int *col4row = reinterpret_cast<int *>(u0 + nmax);
    CT *cT = reinterpret_cast<CT *>(col4row + nmax);
    // CD: the row dual u[i] lives in the padding element at the end of cost row i, so one row scan addresses its three
    // LDS reads from one base with immediate offsets
    double *u = CD ? reinterpret_cast<double *>(cT) + (Qs - 1) : u0;
    const int us = CD ? Qs : 1;

    for (int q = lane; q < Q; q += 64) tgt_for_pred[(long long)p * Q + q] = -1;
    for (int j = lane; j < ldc; j += 64) pred_for_tgt[(long long)p * ldc + j] = -1;
    if (lane == 0) status[p] = 0;
    if (n == 0) return;
    if (n > Q) {
        if (lane == 0) status[p] = 2;
        return;
    }
    const float *Cg = cost + (long long)p * Q * ldc;
    bool bad = false;
    for (int idx = lane; idx < Q * n; idx += 64) {
        const int q = idx / n, j = idx - q * n;
        const float c = Cg[q * ldc + j];
        cT[j * Qs + q] = (CT)c;
        bad |= !(c == c) || (c == -INFINITY);
    }
    if (Qs > Q)                          // columns that do not exist cost +inf: their labels stay +inf without a lane test
        for (int idx = lane; idx < (Qs - Q) * n; idx += 64) cT[(idx / (Qs - Q)) * Qs + Q + idx % (Qs - Q)] = (CT)INFINITY;
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        u[i * us] = 0.0;
        col4row[i] = -1;
    }
    __syncthreads();
    if (__any(bad)) {
        if (lane == 0) status[p] = 1;   // SciPy raises "matrix contains invalid numeric entries"
        return;
    }
    double v[MAXCPL], spc[MAXCPL];
    int path[MAXCPL], row4col[MAXCPL];
    unsigned long long exist[MAXCPL];   // wave-uniform: lanes whose column lane + 64 s exists
    unsigned long long freecol[MAXCPL]; // wave-uniform: existing columns no row is matched to
#pragma unroll
    for (int s = 0; s < MAXCPL; ++s) {
        v[s] = 0.0;
        spc[s] = INFINITY;
        path[s] = -1;
        row4col[s] = -1;
        const int cnt = Q - 64 * s;
        exist[s] = cnt >= 64 ? ~0ull : (cnt <= 0 ? 0ull : ((1ull << cnt) - 1ull));
        freecol[s] = exist[s];
    }
    // Row-reduction start (the classic JV initialisation; SciPy starts from all-zero duals): u[i] = min_j c[i][j] keeps
    // every reduced cost >= 0 with v = 0, and a row whose minimum column is still free is matched to it at reduced
    // cost 0.  About two thirds of the rows of a 99 x 100 problem are settled here; the shortest-augmenting-path
    // phase below only runs for the rest.  The optimum (cost) is unchanged; tie-breaking may differ from SciPy's.
    for (int i = 0; i < n; ++i) {
        const CT *crow = cT + i * Qs;
        double best = INFINITY;
        int bestj = INT_MAX;
#pragma unroll
        for (int s = 0; s < MAXCPL; ++s) {
            const int j = lane + 64 * s;
            if (s < cpl && j < Q) {
                const double c = (double)crow[j];
                if (c < best) { best = c; bestj = j; }
            }
        }
        const double gmin = wave_min_f64_dpp(best);
        if (!(gmin < (double)INFINITY)) continue;          // all-infinite row: left to the search (reports infeasible)
        const unsigned long long mall = __ballot(best == gmin);
        const int ownl = __ffsll((long long)mall) - 1;
        const int jstar = __builtin_amdgcn_readlane(bestj, ownl);
        const int owns = jstar >> 6;
        int r4 = -1;
#pragma unroll
        for (int s = 0; s < MAXCPL; ++s)
            if (s == owns) r4 = row4col[s];
        r4 = __builtin_amdgcn_readlane(r4, ownl);
        if (lane == 0) u[i * us] = gmin;
        if (r4 == -1) {
#pragma unroll
            for (int s = 0; s < MAXCPL; ++s) {
                if (s == owns && lane == ownl) row4col[s] = i;
                if (s == owns) freecol[s] &= ~(1ull << ownl);
            }
            if (lane == 0) col4row[i] = jstar;
        }
    }
    __syncthreads();
#if DETR_ASSIGN_PROF
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_last = __builtin_readcyclecounter();
    const unsigned long long prof_t0 = prof_last;
#endif
    bool infeasible = false;
    for (int cur = 0; cur < n; ++cur) {
        if (col4row[cur] != -1) continue;                  // matched by the row-reduction start
        ASSIGN_TICK(5);
        double minVal = 0.0;
        int i = cur;
        int sink = -1;
        unsigned long long live[MAXCPL], todo[MAXCPL];     // wave-uniform: existing columns not yet scanned; live columns
                                                           // whose label equals minVal
#pragma unroll
        for (int s = 0; s < MAXCPL; ++s) {
            spc[s] = INFINITY;
            live[s] = exist[s];
            todo[s] = 0ull;
        }
        while (true) {
            ASSIGN_TICK(0);
            // scan row i: both LDS reads (u[i] and one cost per slot) are issued together, the relaxations are selects;
            // a column whose new label equals minVal (reduced cost 0) joins `todo`
            const double ui = u[i * us];
            const CT *crow = cT + i * Qs + lane;
            CT cf[MAXCPL];
#pragma unroll
            for (int s = 0; s < MAXCPL; ++s) cf[s] = crow[CD ? 64 * s : 64 * (s < cpl ? s : cpl - 1)];
            unsigned long long any_todo = 0ull;
#pragma unroll
            for (int s = 0; s < MAXCPL; ++s) {
                const double r = minVal + (double)cf[s] - ui - v[s];
                const unsigned long long um = __builtin_amdgcn_ballot_w64(r < spc[s]) & live[s];
                const bool upd = __builtin_amdgcn_inverse_ballot_w64(um);
                spc[s] = upd ? r : spc[s];
                path[s] = upd ? i : path[s];
                todo[s] |= __builtin_amdgcn_ballot_w64(r == minVal) & um;
                any_todo |= todo[s];
            }
            ASSIGN_TICK(1);
            if (any_todo == 0ull) {
                // new level: exact minimum of the live labels, every live column that attains it goes to `todo`
                double best = INFINITY;
#pragma unroll
                for (int s = 0; s < MAXCPL; ++s) {
                    const bool lv = __builtin_amdgcn_inverse_ballot_w64(live[s]);
                    const double c = lv ? spc[s] : (double)INFINITY;
                    asm("v_min_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(c));
                }
                minVal = wave_min_f64_dpp(best);
                // +inf (nothing reachable): scalar test on the high word; the pick below is void then and the loop ends
                infeasible = (__double2hiint(minVal) & 0x7FFFFFFF) >= 0x7FF00000;
#pragma unroll
                for (int s = 0; s < MAXCPL; ++s) todo[s] = __builtin_amdgcn_ballot_w64(spc[s] == minVal) & live[s];
            }
            ASSIGN_TICK(2);
            int jstar = 0, r4 = 0;
            bool isfree = false;
            AssignPick<0, MAXCPL>::run(todo, live, freecol, row4col, jstar, r4, isfree);
            if (isfree | infeasible) {
                sink = jstar;
                break;
            }
            i = r4;
            ASSIGN_TICK(3);
        }
        if (infeasible) break;
        ASSIGN_TICK(4);
        // dual update (before the augmentation, with the old matching)
        if (lane == 0) u[cur * us] += minVal;
#pragma unroll
        for (int s = 0; s < MAXCPL; ++s) {
            if (__builtin_amdgcn_inverse_ballot_w64(exist[s] & ~live[s])) {
                const double d = minVal - spc[s];
                v[s] -= d;
                if (row4col[s] != -1) u[row4col[s] * us] += d;
            }
        }
        __syncthreads();
        // augment along the alternating path that ends in `sink`
#pragma unroll
        for (int s = 0; s < MAXCPL; ++s)
            if (s == (sink >> 6)) freecol[s] &= ~(1ull << (sink & 63));
        int j = sink;
        while (true) {
            const int ownl = j & 63, owns = j >> 6;
            int pi = 0;
#pragma unroll
            for (int s = 0; s < MAXCPL; ++s)
                if (s == owns) pi = path[s];
            pi = __builtin_amdgcn_readlane(pi, __builtin_amdgcn_readfirstlane(ownl));
#pragma unroll
            for (int s = 0; s < MAXCPL; ++s)
                if (s == owns && lane == ownl) row4col[s] = pi;
            const int prev = col4row[pi];
            __syncthreads();
            if (lane == 0) col4row[pi] = j;
            __syncthreads();
            j = prev;
            if (pi == cur) break;
        }
    }
    __syncthreads();
    if (infeasible) {
        if (lane == 0) status[p] = 1;
        return;
    }
    for (int i = lane; i < n; i += 64) {
        const int q = col4row[i];
        pred_for_tgt[(long long)p * ldc + i] = q;
        if (q >= 0) tgt_for_pred[(long long)p * Q + q] = i;
    }
#if DETR_ASSIGN_PROF
    __syncthreads();
    if (lane == 0) {
        for (int k = 0; k < 6; ++k) pred_for_tgt[(long long)p * ldc + k] = (int)prof_acc[k];
        pred_for_tgt[(long long)p * ldc + 6] = (int)(__builtin_readcyclecounter() - prof_t0);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// K14: loss sums, finalisation, gradients.
//   sums[lv*8 + {0: sum w*CE, 1: sum w, 2: neg correct, 3: n_neg, 4: n_pos, 5: pos != bg, 6: pos correct}]
//   sums[levels*8 + lv*2 + {0: sum L1, 1: sum (1-GIoU)}]
// ------------------------------------------------------------------------------------------------
// The cross-entropy normaliser sum(w) (loss.py:60-67: weight 0.1 on the background class, 1 elsewhere) is derived from the
// two COUNTS, which float atomics accumulate exactly in any order, instead of being accumulated itself: 0.1f is not a dyadic
// number, so the order of 4 waves x B images of atomic adds moved it by an ulp from run to run -- and every gradient with
// it (two runs of the same step then differed by up to 1e-2 in bf16-stored activation gradients of the near-degenerate
// random-init model).  With this the gradients are a deterministic function of the matching; only the REPORTED loss values
// (sums of w*CE, L1, 1-GIoU) keep the atomics' last-bit noise.
__device__ __forceinline__ float ce_weight_sum(const float *s) { return 0.1f * s[3] + s[4]; }

__global__ __launch_bounds__(256) void set_loss_sums_kernel(SetLossArgs a, const int *__restrict__ tgt_for_pred,
                                                            float *__restrict__ sums) {
    const int p = blockIdx.x;
    const int lv = p / a.B, b = p % a.B;
    const float *lg = a.logits + lv * a.sL_l + b * a.sL_b;
    const float *bx = a.boxes + lv * a.sB_l + b * a.sB_b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    const int q0 = blockIdx.y * SL_QCHUNK, q1 = min(a.Q, q0 + SL_QCHUNK);
    for (int q = q0 + wave; q < q1; q += 4) {
        const float *row = lg + q * a.sL_q;
        const int t = tgt_for_pred[(long long)p * a.Q + q];
        // logsumexp + first-occurrence argmax over the classes
        float mx = -INFINITY;
        int am = INT_MAX;
        for (int c = lane; c < a.C; c += 64) {
            const float x = row[c];
            if (x > mx) { mx = x; am = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(mx, o, 64);
            const int oa = __shfl_xor(am, o, 64);
            if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
        }
        float s = 0.f;
        for (int c = lane; c < a.C; c += 64) s += expf(row[c] - mx);
        s = wave_sum(s);
        if (lane == 0) {
            const float lse = logf(s) + mx;
            if (t < 0) {
                const float ce = lse - row[a.background_class];
                acc[0] += 0.1f * ce;
                acc[2] += (am == a.background_class) ? 1.f : 0.f;
                acc[3] += 1.f;
            } else {
                const int cls = (int)a.t_class[(long long)b * a.R + 1 + t];
                // a label outside [0, C) (e.g. the wrong nb_class when finetuning): no out-of-bounds read; the loss turns
                // NaN like tf.nn.sparse_softmax_cross_entropy_with_logits on the GPU (the CPU op raises)
                const float ce = (cls >= 0 && cls < a.C) ? lse - row[cls] : NAN;
                acc[0] += ce;
                acc[4] += 1.f;
                acc[5] += (am != a.background_class) ? 1.f : 0.f;
                acc[6] += (am == cls) ? 1.f : 0.f;
                const float *pb = bx + q * a.sB_q;
                const float *tb = a.t_bbox + ((long long)b * a.R + 1 + t) * 4;
                acc[7] += fabsf(pb[0] - tb[0]) + fabsf(pb[1] - tb[1]) + fabsf(pb[2] - tb[2]) + fabsf(pb[3] - tb[3]);
                acc[8] += 1.0f - giou_of(to_xyxy(pb[0], pb[1], pb[2], pb[3]), to_xyxy(tb[0], tb[1], tb[2], tb[3]));
            }
        }
    }
    // the four waves' partials are combined in LDS first: one set of atomics per workgroup (with a workgroup per query chunk
    // there are 13x as many workgroups adding into the same 60 addresses)
    __shared__ float red[4][9];
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const int k = threadIdx.x;
        const float v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        if (v != 0.f) unsafeAtomicAdd(k < 7 ? sums + lv * 8 + k : sums + a.levels * 8 + lv * 2 + (k - 7), v);
    }
}

__global__ void set_loss_finalize_kernel(const float *__restrict__ sums, int levels, float *__restrict__ losses,
                                         float *__restrict__ total) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float tot = 0.f;
    // the reference sums the main output first (last decoder level), then aux 0..levels-2 (loss.py:22-34)
    for (int k = 0; k < levels; ++k) {
        const int lv = (k == 0) ? levels - 1 : k - 1;
        const float *s = sums + lv * 8;
        const float *sb = sums + levels * 8 + lv * 2;
        const float label = s[0] / ce_weight_sum(s);
        const float npos = s[4];
        const float giou = sb[1] / npos;
        const float l1 = sb[0] / npos;
        float *o = losses + lv * 6;
        o[0] = label;
        o[1] = s[2] / s[3];
        o[2] = s[5] / npos;
        o[3] = s[6] / npos;
        o[4] = giou;
        o[5] = l1;
        tot += label * 1.0f;
        tot += giou * 2.0f;
        tot += l1 * 5.0f;
    }
    total[0] = tot;
}

__global__ __launch_bounds__(256) void set_loss_grad_kernel(SetLossArgs a, const int *__restrict__ tgt_for_pred,
                                                            const float *__restrict__ sums, float loss_scale,
                                                            float *__restrict__ d_logits, float *__restrict__ d_boxes) {
    const int p = blockIdx.x;
    const int lv = p / a.B, b = p % a.B;
    const float *lg = a.logits + lv * a.sL_l + b * a.sL_b;
    const float *bx = a.boxes + lv * a.sB_l + b * a.sB_b;
    float *dlg = d_logits + lv * a.sL_l + b * a.sL_b;
    float *dbx = d_boxes + lv * a.sB_l + b * a.sB_b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float sw = ce_weight_sum(sums + lv * 8);
    const float npos = sums[lv * 8 + 4];
    const int q0 = blockIdx.y * SL_QCHUNK, q1 = min(a.Q, q0 + SL_QCHUNK);
    for (int q = q0 + wave; q < q1; q += 4) {
        const float *row = lg + q * a.sL_q;
        const int t = tgt_for_pred[(long long)p * a.Q + q];
        float mx = -INFINITY;
        for (int c = lane; c < a.C; c += 64) mx = fmaxf(mx, row[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int c = lane; c < a.C; c += 64) s += expf(row[c] - mx);
        s = wave_sum(s);
        const int cls = (t < 0) ? a.background_class : (int)a.t_class[(long long)b * a.R + 1 + t];
        const float wq = ((t < 0) ? 0.1f : 1.0f) / sw * loss_scale;   // label weight 1 (loss.py:10-11)
        for (int c = lane; c < a.C; c += 64) {
            const float sm = expf(row[c] - mx) / s;
            dlg[q * a.sL_q + c] = wq * (sm - (c == cls ? 1.0f : 0.0f));
        }
        if (lane == 0) {
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            if (t >= 0) {
                const float *pb = bx + q * a.sB_q;
                const float *tb = a.t_bbox + ((long long)b * a.R + 1 + t) * 4;
                const float kl1 = 5.0f * loss_scale / npos;
                const float kg = 2.0f * loss_scale / npos;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = pb[k] - tb[k];
                    g[k] = kl1 * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
                }
                // d(1 - giou)/d(cx,cy,w,h) through the clipped corners
                const float ax1 = pb[0] - pb[2] / 2.0f, ay1 = pb[1] - pb[3] / 2.0f;
                const float ax2 = pb[0] + pb[2] / 2.0f, ay2 = pb[1] + pb[3] / 2.0f;
                const float mx1 = (ax1 >= 0.f && ax1 <= 1.f) ? 1.f : 0.f, my1 = (ay1 >= 0.f && ay1 <= 1.f) ? 1.f : 0.f;
                const float mx2 = (ax2 >= 0.f && ax2 <= 1.f) ? 1.f : 0.f, my2 = (ay2 >= 0.f && ay2 <= 1.f) ? 1.f : 0.f;
                const XYXY P = to_xyxy(pb[0], pb[1], pb[2], pb[3]);
                const XYXY T = to_xyxy(tb[0], tb[1], tb[2], tb[3]);
                const float iwr = fminf(P.x2, T.x2) - fmaxf(P.x1, T.x1), ihr = fminf(P.y2, T.y2) - fmaxf(P.y1, T.y1);
                const float iw = fmaxf(iwr, 0.f), ih = fmaxf(ihr, 0.f);
                const float inter = iw * ih;
                const float pw_ = P.x2 - P.x1, ph_ = P.y2 - P.y1;
                const float U = pw_ * ph_ + (T.x2 - T.x1) * (T.y2 - T.y1) - inter;
                const float cwr = fmaxf(P.x2, T.x2) - fminf(P.x1, T.x1), chr_ = fmaxf(P.y2, T.y2) - fminf(P.y1, T.y1);
                const float cw = fmaxf(cwr, 0.f), ch = fmaxf(chr_, 0.f);
                const float H = cw * ch;
                // partials of iw, ih, cw, ch w.r.t. the four corners (TF min/max tie convention: first arg on ties)
                const float diw_x2 = (iwr > 0.f && P.x2 <= T.x2) ? 1.f : 0.f;
                const float diw_x1 = (iwr > 0.f && P.x1 >= T.x1) ? -1.f : 0.f;
                const float dih_y2 = (ihr > 0.f && P.y2 <= T.y2) ? 1.f : 0.f;
                const float dih_y1 = (ihr > 0.f && P.y1 >= T.y1) ? -1.f : 0.f;
                const float dcw_x2 = (cwr > 0.f && P.x2 >= T.x2) ? 1.f : 0.f;
                const float dcw_x1 = (cwr > 0.f && P.x1 <= T.x1) ? -1.f : 0.f;
                const float dch_y2 = (chr_ > 0.f && P.y2 >= T.y2) ? 1.f : 0.f;
                const float dch_y1 = (chr_ > 0.f && P.y1 <= T.y1) ? -1.f : 0.f;
                const float dI[4] = {ih * diw_x1, iw * dih_y1, ih * diw_x2, iw * dih_y2};   // x1,y1,x2,y2
                const float dA[4] = {-ph_, -pw_, ph_, pw_};
                const float dH[4] = {ch * dcw_x1, cw * dch_y1, ch * dcw_x2, cw * dch_y2};
                float gc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dU = dA[k] - dI[k];
                    const float diou = dI[k] / U - inter * dU / (U * U);
                    const float dgiou = diou - (U * dH[k] - H * dU) / (H * H);
                    gc[k] = -kg * dgiou;
                }
                g[0] += gc[0] * mx1 + gc[2] * mx2;
                g[1] += gc[1] * my1 + gc[3] * my2;
                g[2] += 0.5f * (gc[2] * mx2 - gc[0] * mx1);
                g[3] += 0.5f * (gc[3] * my2 - gc[1] * my1);
            }
            float *o = dbx + q * a.sB_q;
            o[0] = g[0]; o[1] = g[1]; o[2] = g[2]; o[3] = g[3];
        }
    }
}

static int fill_args(const detr_setloss_desc *d, SetLossArgs &a) {
    DETR_REQUIRE(d != nullptr, "setloss: null descriptor");
    DETR_REQUIRE(d->levels > 0 && d->B > 0 && d->Q > 0 && d->C > 0 && d->R > 1, "setloss: bad shape");
    DETR_REQUIRE(d->Q <= SL_MAXQ && d->R <= SL_MAXR, "setloss: Q=%d (max %d) R=%d (max %d)", d->Q, SL_MAXQ, d->R, SL_MAXR);
    DETR_REQUIRE(d->logits && d->boxes && d->t_bbox && d->t_class, "setloss: null operand");
    DETR_REQUIRE(d->background_class >= 0 && d->background_class < d->C, "setloss: background_class out of range");
    a.levels = d->levels; a.B = d->B; a.Q = d->Q; a.C = d->C; a.R = d->R;
    a.logits = d->logits; a.sL_l = d->sL_l; a.sL_b = d->sL_b; a.sL_q = d->sL_q;
    a.boxes = d->boxes; a.sB_l = d->sB_l; a.sB_b = d->sB_b; a.sB_q = d->sB_q;
    a.t_bbox = d->t_bbox; a.t_class = reinterpret_cast<const long long *>(d->t_class);
    a.background_class = d->background_class;
    return 0;
}

}  // namespace detr

using namespace detr;

extern "C" int detr_hip_match_cost_f32(const detr_setloss_desc *d, float *cost, void *stream) {
    SetLossArgs a;
    if (fill_args(d, a)) return -1;
    DETR_REQUIRE(cost, "match_cost: null cost");
    hipLaunchKernelGGL(match_cost_kernel, dim3(a.levels * a.B, cdiv(a.Q, SL_QCHUNK)), dim3(256), 0, (hipStream_t)stream, a, cost);
    DETR_LAUNCH_CHECK("match_cost");
    return 0;
}

extern "C" int detr_hip_assign_f32(const float *cost, int32_t P, int32_t Q, int32_t ldc, const float *t_bbox, int32_t B,
                                   int32_t R, int32_t *tgt_for_pred, int32_t *pred_for_tgt, int32_t *status,
                                   void *stream) {
    DETR_REQUIRE(cost && t_bbox && tgt_for_pred && pred_for_tgt && status, "assign: null operand");
    DETR_REQUIRE(P > 0 && Q > 0 && Q <= SL_MAXQ && R > 1 && R <= SL_MAXR && ldc >= R - 1 && B > 0 && P % B == 0,
                 "assign: bad shape P=%d Q=%d ldc=%d B=%d R=%d", P, Q, ldc, B, R);
    const int nmax = (R + 1) & ~1;
    const bool cd = Q <= 128 && (size_t)nmax * 12 + (size_t)(R - 1) * 129 * 8 <= 150 * 1024;   // cost tile as doubles, row stride 129
    const size_t tile = (size_t)(R - 1) * (size_t)(cd ? 129 : (Q + 63) / 64 * 64 + 1);
    const size_t smem = (size_t)nmax * 8 + (size_t)nmax * 4 + tile * (cd ? 8 : 4);
    hipStream_t s = (hipStream_t)stream;
    const int cpl = (Q + 63) / 64;
#define DETR_ASSIGN_LAUNCH(MC, CD)                                                                                   \
    do {                                                                                                             \
        if (smem > 48 * 1024) {                                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(assign_kernel<MC, CD>),                \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);               \
            DETR_REQUIRE(e == hipSuccess, "assign: cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e)); \
        }                                                                                                            \
        hipLaunchKernelGGL((assign_kernel<MC, CD>), dim3(P), dim3(64), smem, s, cost, Q, ldc, t_bbox, B, R,          \
                           tgt_for_pred, pred_for_tgt, status);                                                      \
    } while (0)
    if (cpl <= 2 && cd) DETR_ASSIGN_LAUNCH(2, true);
    else if (cpl <= 2) DETR_ASSIGN_LAUNCH(2, false);
    else if (cpl <= 5) DETR_ASSIGN_LAUNCH(5, false);
    else DETR_ASSIGN_LAUNCH(8, false);
#undef DETR_ASSIGN_LAUNCH
    DETR_LAUNCH_CHECK("assign");
    return 0;
}

extern "C" int detr_hip_set_loss_sums_f32(const detr_setloss_desc *d, const int32_t *tgt_for_pred, float *sums,
                                          void *stream) {
    SetLossArgs a;
    if (fill_args(d, a)) return -1;
    DETR_REQUIRE(tgt_for_pred && sums, "set_loss_sums: null operand");
    hipLaunchKernelGGL(set_loss_sums_kernel, dim3(a.levels * a.B, cdiv(a.Q, SL_QCHUNK)), dim3(256), 0, (hipStream_t)stream, a, tgt_for_pred,
                       sums);
    DETR_LAUNCH_CHECK("set_loss_sums");
    return 0;
}

extern "C" int detr_hip_set_loss_finalize_f32(const float *sums, int32_t levels, float *losses, float *total,
                                              void *stream) {
    DETR_REQUIRE(sums && losses && total && levels > 0, "set_loss_finalize: bad args");
    hipLaunchKernelGGL(set_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, levels, losses, total);
    DETR_LAUNCH_CHECK("set_loss_finalize");
    return 0;
}

extern "C" int detr_hip_set_loss_grad_f32(const detr_setloss_desc *d, const int32_t *tgt_for_pred, const float *sums,
                                          float loss_scale, float *d_logits, float *d_boxes, void *stream) {
    SetLossArgs a;
    if (fill_args(d, a)) return -1;
    DETR_REQUIRE(tgt_for_pred && sums && d_logits && d_boxes, "set_loss_grad: null operand");
    hipLaunchKernelGGL(set_loss_grad_kernel, dim3(a.levels * a.B, cdiv(a.Q, SL_QCHUNK)), dim3(256), 0, (hipStream_t)stream, a, tgt_for_pred,
                       sums, loss_scale, d_logits, d_boxes);
    DETR_LAUNCH_CHECK("set_loss_grad");
    return 0;
}
