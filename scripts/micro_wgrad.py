"""Cold-cache sweep of the split-K weight-gradient GEMMs (C[M,N] += A[K,M]^T B[K,N], bf16 operands, fp32 slabs): tile size
(DETR_HIP_GEMM_TILE) x split count, each launch timed on its own after a 600 MB flush of L2 / Infinity Cache -- back-to-back
launches of one shape run out of the 256 MB Infinity Cache and flatter every variant.  Time = GEMM + its slab reduction.
usage: python scripts/micro_wgrad.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    sys.path.insert(0, p)
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
hip.ensure_workspace(dev)
hip.COMPUTE_BF16 = 1
bf = torch.bfloat16
flush = torch.empty(600 * 1024 * 1024 // 4, device=dev)
shapes = [(1024, 512, 33600), (256, 1024, 33600), (1024, 256, 33600), (128, 512, 133600), (512, 128, 133600), (64, 256, 534400),
          (256, 64, 534400), (256, 128, 534400), (512, 2048, 8400), (2048, 512, 8400), (1024, 2048, 8400), (256, 2048, 8400),
          (512, 1024, 33600), (256, 512, 133600)]
rows = []
for M, N, K in shapes:
    torch.manual_seed(0)
    A = torch.randn(K, M, device=dev).to(bf)
    B = torch.randn(K, N, device=dev).to(bf)
    C = torch.zeros(M, N, device=dev)
    base = hip.pick_split_k(M, N, K)
    best = None
    line = []
    for tile, tname in ((0, "auto"), (1, "128x128"), (5, "64x128"), (3, "64x64")):
        for sk in sorted({max(1, base // 2), base, base * 2, base * 4}):
            if sk > 1024:
                continue
            hip.set_tuning("DETR_HIP_GEMM_TILE", tile if tile else None)
            try:
                ts = []
                for rep in range(7):
                    flush.zero_()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    hip.gemm(M, N, K, A, M, 0, B, N, 0, C, N, compute=1, split_k=sk)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                t = sorted(ts)[len(ts) // 2]
            finally:
                hip.set_tuning("DETR_HIP_GEMM_TILE", None)
            line.append((tname, sk, t))
            if best is None or t < best[2]:
                best = (tname, sk, t)
    cur = [x for x in line if x[0] == "auto" and x[1] == base][0]
    print(f"M{M} N{N} K{K}: current auto sk{base} {cur[2]:.1f} us; best {best[0]} sk{best[1]} {best[2]:.1f} us | " +
          " ".join(f"{n}/sk{s}:{t:.0f}" for n, s, t in line))
    rows.append(dict(M=M, N=N, K=K, base_split=base, current_us=cur[2], best=best, all=line))
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
