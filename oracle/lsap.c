/* ORACLE (test infrastructure only) -- plain-C restatement of the rectangular linear sum
 * assignment solver the reference calls through
 *   scipy.optimize.linear_sum_assignment   (detr_tf/loss/hungarian_matching.py:7,29).
 *
 * The algorithm lives in SciPy (third-party, NOT under /root/reference; the reference does
 * not pin a version, this image ships SciPy 1.15.3).  SciPy's solver is the modified
 * Jonker-Volgenant shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D
 * rectangular assignment algorithms", IEEE TAES 52(4), 2016; this file restates that
 * published algorithm (dual variables u, v; one Dijkstra-like augmentation per row; the
 * cost matrix is transposed when it has more rows than columns; double arithmetic).
 * tests/test_oracle_lsap.py pins it against the real SciPy on random, tied and ragged
 * matrices.  Built by __graft_entry__.build() into oracle/_build/liblsap_oracle.so; used
 * only as a checker and as the cpu_baseline of the matcher.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* returns 0 ok, -1 infeasible, -2 invalid (NaN / -inf) ; a,b have min(nr,nc) entries */
int lsap_oracle_solve(int nr, int nc, const double *cost_in, int64_t *a, int64_t *b)
{
    if (nr == 0 || nc == 0) return 0;
    int transpose = nc < nr;
    double *cost = (double *)malloc(sizeof(double) * (size_t)nr * nc);
    if (transpose) {
        for (int i = 0; i < nr; i++)
            for (int j = 0; j < nc; j++) cost[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
        int t = nr; nr = nc; nc = t;
    } else {
        memcpy(cost, cost_in, sizeof(double) * (size_t)nr * nc);
    }
    for (size_t i = 0; i < (size_t)nr * nc; i++)
        if (cost[i] != cost[i] || cost[i] == -INFINITY) { free(cost); return -2; }

    double *u = (double *)calloc(nr, sizeof(double));
    double *v = (double *)calloc(nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * nc);     /* shortest path costs */
    int *path = (int *)malloc(sizeof(int) * nc);
    int *col4row = (int *)malloc(sizeof(int) * nr);
    int *row4col = (int *)malloc(sizeof(int) * nc);
    char *SR = (char *)malloc(nr), *SC = (char *)malloc(nc);
    int *remaining = (int *)malloc(sizeof(int) * nc);
    for (int i = 0; i < nr; i++) col4row[i] = -1;
    for (int j = 0; j < nc; j++) { row4col[j] = -1; path[j] = -1; }
    int rc = 0;

    for (int cur = 0; cur < nr && rc == 0; cur++) {
        double minVal = 0.0;
        int i = cur, num_remaining = nc, sink = -1;
        for (int it = 0; it < nc; it++) remaining[it] = nc - it - 1;
        memset(SR, 0, nr); memset(SC, 0, nc);
        for (int j = 0; j < nc; j++) spc[j] = INFINITY;
        while (sink == -1) {
            int index = -1; double lowest = INFINITY;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; it++) {
                int j = remaining[it];
                double r = minVal + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (minVal == INFINITY) { rc = -1; break; }
            int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        if (rc) break;
        u[cur] += minVal;
        for (int r = 0; r < nr; r++) if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; j++) if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        for (;;) {
            int r = path[j];
            row4col[j] = r;
            int t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
    if (rc == 0) {
        if (transpose) {
            /* rows of the transposed problem are the caller's columns: emit sorted by caller row */
            int k = 0;
            for (int j = 0; j < nc; j++) if (row4col[j] != -1) { a[k] = j; b[k] = row4col[j]; k++; }
        } else {
            for (int r = 0; r < nr; r++) { a[r] = r; b[r] = col4row[r]; }
        }
    }
    free(cost); free(u); free(v); free(spc); free(path); free(col4row); free(row4col);
    free(SR); free(SC); free(remaining);
    return rc;
}
