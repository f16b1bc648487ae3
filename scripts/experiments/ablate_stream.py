"""Times the streaming short-K GEMM (csrc/gemm_stream.h) on its shapes of the B=8 800x1333 step with the library build
DETR_HIP_LIB points at (ablation builds: scripts/experiments/ablate.sh with ABLATE_FILES=gemm_f32).  One line per shape."""
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


def timeit(fn, reps=10, warm=3):
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
out = []
for (M, N, K, bk, res, mask, act) in [(534400, 256, 64, 0, 1, 0, 1), (534400, 256, 64, 1, 1, 1, 0), (534400, 64, 256, 1, 0, 1, 0),
                                      (133600, 512, 128, 0, 1, 0, 1), (133600, 512, 128, 1, 1, 1, 0),
                                      (33600, 1024, 256, 0, 1, 0, 1), (33600, 1024, 256, 1, 1, 1, 0), (133600, 512, 256, 1, 1, 1, 0), (8400, 2048, 256, 1, 0, 0, 1), (8400, 2048, 256, 0, 0, 1, 0)]:
    A, Bm, C = bf(M, K), (bf(N, K) if bk else bf(K, N)), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    R, Mk, bias = (bf(M, N) if res else None), (bf(M, N) if mask else None), torch.randn(N, device=dev)
    nbytes = 2 * (M * K + K * N + M * N * (1 + res + mask))
    us = timeit(lambda: hip.gemm(M, N, K, A, K, 1, Bm, K if bk else N, bk, C, N, bias=bias, residual=R, ldr=N if res else 0,
                                 mask=Mk, ldmask=N if mask else 0, act=act, compute=1))
    out.append(f"M{M}_N{N}_K{K}_r{res}m{mask} {us:.1f}us {nbytes / us / 1e6:.2f}TB/s")
print(" | ".join(out), flush=True)
