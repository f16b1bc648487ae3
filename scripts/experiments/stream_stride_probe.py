import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/detr-tensorflow_amd")
import torch
from detr_tf import _hip as hip
hip.load(); dev = "cuda"; hip.ensure_workspace(dev)
def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
M, K = 534400, 64
for N, ld in [(64, 64), (64, 128), (64, 256), (64, 512), (128, 128), (128, 512), (256, 256)]:
    A, Bm = bf(M, K), bf(K, N)
    C, R = torch.empty(M, ld, device=dev, dtype=torch.bfloat16), bf(M, ld)
    bias = torch.randn(N, device=dev)
    nbytes = 2 * (M * K + K * N + 2 * M * N)
    us = timeit(lambda: hip.gemm(M, N, K, A, K, 1, Bm, N, 0, C, ld, bias=bias, residual=R, ldr=ld, act=1, compute=1))
    print(f"M{M} N{N} K{K} ldc{ld}: {us:7.1f} us  {nbytes/us/1e6:5.2f} TB/s", flush=True)
