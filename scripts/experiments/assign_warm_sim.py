"""Matcher: would warm-started duals across decoder levels cut the scanned rows?  (CPU simulation, no GPU.)

    python scripts/experiments/assign_warm_sim.py COST.npz     COST.npz from `assign_probe.py dump` ([48, 100, 99] cost tensors of
                                                               bench steps: 6 levels x 8 images, + t_bbox with the target counts)

The shortest-augmenting-path solver below is SciPy's / the kernel's algorithm (rows = targets, columns = queries; `steps` = scanned rows,
the unit the kernel's time is made of: ~640 cycles each).  Warm start for level l+1: column prices v of level l, u_i = min_j(c_ij - v_j),
empty matching.  Two findings (NOTEBOOK 4c):
  * the transferred prices are not dual-feasible for a RECTANGULAR problem: a column that ends unmatched needs v_j = 0, the previous level's
    price is generally < 0 there -- the warm-started solver returns assignments that are NOT optimal (column `opt gap` > 0) unless the problem is
    squared with dummy rows;
  * even ignoring that, a random-init model's levels are not close: the n = 99 problems scan MORE rows warm than cold."""
import sys

import numpy as np
from scipy.optimize import linear_sum_assignment


def solve(cost, v0=None):
    C = cost.T.astype(np.float64)                     # [n targets, Q queries]
    n, Q = C.shape
    col4row = -np.ones(n, int)
    row4col = -np.ones(Q, int)
    v = np.zeros(Q) if v0 is None else v0.copy()
    u = np.zeros(n) if v0 is None else (C - v[None, :]).min(axis=1)
    steps = 0
    for cur in range(n):
        min_val, i = 0.0, cur
        spc = np.full(Q, np.inf)
        path = -np.ones(Q, int)
        SC = np.zeros(Q, bool)
        while True:
            steps += 1
            r = min_val + C[i] - u[i] - v
            upd = (~SC) & (r < spc)
            spc[upd] = r[upd]
            path[upd] = i
            cand = np.where(~SC, spc, np.inf)
            m = cand.min()
            js = np.where(cand == m)[0]
            free = [j for j in js if row4col[j] == -1]
            j = free[0] if free else js[0]
            min_val = m
            SC[j] = True
            if row4col[j] == -1:
                sink = j
                break
            i = row4col[j]
        u[cur] += min_val
        for j in np.where(SC)[0]:
            d = min_val - spc[j]
            v[j] -= d
            if row4col[j] != -1 and j != sink:
                u[row4col[j]] += d
        j = sink
        while True:
            pi = path[j]
            row4col[j] = pi
            col4row[pi], j = j, col4row[pi]
            if pi == cur:
                break
    return col4row, v, steps


def main(path):
    d = np.load(path)
    keys = [k for k in d.files if k.startswith("cost")]
    print(f"{'tensor':8s} {'img':>3s} {'n':>3s} {'lvl':>3s} {'cold steps':>10s} {'warm steps':>10s} {'opt gap (warm)':>14s}")
    tot_c = tot_w = 0
    for key in keys:
        for img in range(8):
            n = int(d["t_bbox"][img, 0, 0]) if d["t_bbox"].ndim == 3 and d["t_bbox"][img, 0, 0] >= 1 else None
            if n is None:
                n = int((np.abs(d["t_bbox"][img]).sum(axis=-1) > 0).sum())
            n = max(1, min(n, 99))
            v = None
            for lv in range(6):
                c = d[key][lv * 8 + img][:, :n]
                _, _, s_cold = solve(c)
                c4r, v, s_warm = solve(c, v)
                rows, cols = linear_sum_assignment(c)
                gap = c[c4r, np.arange(n)].astype(np.float64).sum() - c[rows, cols].astype(np.float64).sum()
                if lv > 0:
                    tot_c += s_cold
                    tot_w += s_warm
                if n >= 50 or lv == 5:
                    print(f"{key:8s} {img:3d} {n:3d} {lv:3d} {s_cold:10d} {s_warm:10d} {gap:14.3e}")
    print(f"levels 1-5, all images: cold {tot_c} scanned rows, warm {tot_w}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "scratch/cost.npz")
