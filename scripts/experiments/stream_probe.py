"""Per-strip section times of the streaming GEMM (csrc/gemm_stream.h built with -DDETR_STREAM_PROF=1, library given by DETR_HIP_LIB):
s_memtime ticks accumulated per wave, written over the maskbits_out buffer.  One line per shape: median over waves of the section
sums divided by the wave's strip count."""
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
bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
NAMES = ["requests", "chunk0", "chunk1", "staged", "items", "t0", "loop", "between"]
# (the probe rides on the mask-bits-out form of the kernel: [k][n] weights + residual, see gemm_stream_eligible)
for (M, N, K, bk, res, act) in [(33600, 1024, 256, 0, 1, 1), (8400, 2048, 256, 0, 1, 1), (133600, 512, 128, 0, 1, 1), (534400, 256, 64, 0, 1, 1)]:
    A, Bm, C = bf(M, K), (bf(N, K) if bk else bf(K, N)), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    R, bias = (bf(M, N) if res else None), torch.randn(N, device=dev)
    prof = torch.zeros(M, N // 8, device=dev, dtype=torch.uint8)
    run = lambda: hip.gemm(M, N, K, A, K, 1, Bm, K if bk else N, bk, C, N, bias=bias, residual=R, ldr=N if res else 0, act=act, compute=1,
                           maskbits_out=prof)
    for _ in range(3):
        run()
    prof.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    t = prof.view(-1).view(torch.int64).view(-1, 8).cpu()
    t = t[t[:, 6] > 0]
    loop = t[:, 6].double()
    t0 = t[:, 5].double()
    secs = t[:, [0, 1, 2, 3, 4, 7]].double()
    print(f"M{M} N{N} K{K} res{res}: {e0.elapsed_time(e1) * 1e3:.1f} us, {t.shape[0]} waves; loop ticks median {loop.median():.0f} max {loop.max():.0f}; "
          f"start spread {(t0.max() - t0.min()):.0f}; end spread {((t0 + loop).max() - (t0 + loop).min()):.0f}; whole {(t0 + loop).max() - t0.min():.0f}")
    print("   section share of loop (median over waves): " + ", ".join(f"{n} {float((secs[:, i] / loop).median()):.3f}" for i, n in enumerate(["requests", "chunk0", "chunk1", "staged", "items", "between"])))
