"""Times the bf16 conv3x3 / GEMM tile kernels on the shapes of the B=8 800x1333 step with whatever build of the library
DETR_HIP_LIB points at (scripts/experiments/ablate.sh runs it once per -DDETR_ABLATE=<bits> build; the ablated builds
compute garbage -- only their time is of interest).  Prints one line per shape: name, microseconds."""
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
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)


def timeit(fn, reps=8, warm=2):
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


out = []
for (N, H, W, C) in [(8, 200, 334, 64), (8, 100, 167, 128), (8, 50, 84, 256), (8, 25, 42, 512)]:
    x, w, y = bf(N, H, W, C), bf(3, 3, C, C), torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(C, device=dev)
    for mode, nm in ((0, "fwd"), (1, "dgrad")):
        us = timeit(lambda: hip.conv3x3(mode, x, w, y, N, H, W, C, H, W, C, 1, bias=bias if mode == 0 else None, act=1 if mode == 0 else 0,
                                        mask=None, compute=1))
        out.append((f"conv3x3_{nm}_{H}x{W}x{C}", us, 2.0 * N * H * W * 9 * C * C))
for (N, H, W, C) in [(8, 200, 334, 64), (8, 100, 167, 128), (8, 50, 84, 256), (8, 25, 42, 512)]:
    x, dy, dw = bf(N, H, W, C), bf(N, H, W, C), torch.zeros(3, 3, C, C, device=dev)
    us = timeit(lambda: hip.conv3x3(2, x, dy, dw, N, H, W, C, H, W, C, 1, compute=1))
    out.append((f"conv3x3_wgrad_{H}x{W}x{C}", us, 2.0 * N * H * W * 9 * C * C))
for (M, Nn, K, ak, bk, sk) in [(33600, 256, 1024, 1, 1, 1), (8400, 2048, 256, 1, 0, 1), (8400, 256, 2048, 1, 1, 1), (133600, 128, 512, 1, 1, 1),
                               (256, 1024, 33600, 0, 0, 32), (64, 256, 534400, 0, 0, 256), (1024, 256, 33600, 0, 0, 32)]:
    A = bf(M, K) if ak else bf(K, M)
    Bm = bf(Nn, K) if bk else bf(K, Nn)
    C = torch.zeros(M, Nn, device=dev, dtype=torch.float32 if sk > 1 else torch.bfloat16)
    us = timeit(lambda: hip.gemm(M, Nn, K, A, K if ak else M, ak, Bm, K if bk else Nn, bk, C, Nn, split_k=sk, compute=1))
    out.append((f"gemm_M{M}_N{Nn}_K{K}_ak{ak}bk{bk}sk{sk}", us, 2.0 * M * Nn * K))
print(" | ".join(f"{n} {us:.1f}us {fl / us / 1e6:.0f}TF" for n, us, fl in out), flush=True)
