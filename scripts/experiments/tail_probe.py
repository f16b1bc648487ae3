"""Tile-count quantisation of the bf16 tile engine: M33600 N256 K1024 is 526 tiles of 128x128 on 512 resident slots (2 workgroups per CU).
Times the same GEMM at row counts around the slot boundary.  One line per M."""
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
for (N, K, bk) in [(256, 1024, 1), (256, 1024, 0), (128, 512, 1), (256, 2048, 1)]:
    out = []
    for M in ([16384, 24576, 32768, 33600, 34816, 49152, 65536] if N == 256 and K == 1024 else
              [131072, 133600, 139264] if N == 128 else [8192, 8400, 12288, 16384]):
        A, Bm, C = bf(M, K), (bf(N, K) if bk else bf(K, N)), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: hip.gemm(M, N, K, A, K, 1, Bm, K if bk else N, bk, C, N, compute=1))
        tiles = -(-M // 128) * -(-N // 128)
        out.append(f"M{M} ({tiles} tiles) {us:.1f}us {2 * M * N * K / us / 1e6:.0f}TF")
    print(f"N{N} K{K} bk{bk}: " + " | ".join(out), flush=True)
