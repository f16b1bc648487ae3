"""Per-shape tile sweep of the unsplit bf16 tile-engine GEMMs of the step: DETR_HIP_GEMM_TILE 0 (rule) / 1 (128x128) / 2 (128x64) / 3 (64x64) /
5 (64x128), all-bf16 operands and output.  One line per shape, microseconds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
hip.ensure_workspace(dev)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
SHAPES = [(33600, 256, 1024, 1, 0, 1), (33600, 256, 1024, 0, 0, 0), (8400, 256, 2048, 1, 1, 0), (8400, 256, 2048, 0, 1, 0), (133600, 128, 512, 1, 0, 1),
          (133600, 128, 512, 0, 0, 0), (8400, 2048, 512, 0, 1, 0), (8400, 512, 2048, 1, 0, 1), (33600, 1024, 512, 1, 1, 1), (133600, 256, 512, 0, 0, 0),
          (8400, 256, 768, 0, 1, 0)]
for (M, N, K, bk, res, mask) in SHAPES:
    A, Bm, C = bf(M, K), (bf(N, K) if bk else bf(K, N)), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    R, Mk = (bf(M, N) if res else None), (bf(M, N) if mask else None)
    out = []
    for t in ("0", "1", "2", "3", "5"):
        hip.set_tuning("DETR_HIP_GEMM_TILE", t)
        us = timeit(lambda: hip.gemm(M, N, K, A, K, 1, Bm, K if bk else N, bk, C, N, residual=R, ldr=N if res else 0, mask=Mk, ldmask=N if mask else 0,
                                     compute=1))
        out.append(f"{t}:{us:.1f}")
    hip.set_tuning("DETR_HIP_GEMM_TILE", None)
    print(f"M{M} N{N} K{K} bk{bk} res{res} mask{mask}: " + "  ".join(out), flush=True)
