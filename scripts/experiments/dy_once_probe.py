"""Does the memory-side cache (256 MB) serve the second read of dY?  The bottleneck conv3 backward of layer1 reads g [M, 256] twice:
input gradient dz2 = mask(g @ W3^T) and weight gradient dW3 = y2^T @ g.  Here the pair runs (a) as two launches over all M rows and
(b) interleaved over row chunks, (c) as the fused kernel of csrc/bwd_fused.hip, and (input gradient of chunk i, weight gradient of chunk i, accumulating): if the second read of a 34-68 MB
chunk comes out of the cache, (b) is faster.  usage: python scripts/experiments/dy_once_probe.py"""
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
hip.COMPUTE_BF16 = 1
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
FUSED_ONLY = os.environ.get("PROBE_FUSED_ONLY") == "1"
for (M, d1, d2) in ([(534400, 64, 256)] if FUSED_ONLY else [(534400, 64, 256), (133600, 128, 512)]):
    g, y2, w3 = bf(M, d2), bf(M, d1), bf(d1, d2)
    dz2 = torch.empty(M, d1, device=dev, dtype=torch.bfloat16)
    G = torch.zeros(d1, d2, device=dev)
    scale = torch.ones(d2, device=dev)

    def pair(r0, r1):
        m = r1 - r0
        hip.gemm(m, d1, d2, g[r0:r1], d2, 1, w3, d2, 1, dz2[r0:r1], d1, mask=y2[r0:r1], ldmask=d1)
        sk = hip.pick_split_k(d1, d2, m)
        hip.gemm(d1, d2, m, y2[r0:r1], d1, 0, g[r0:r1], d2, 0, G, d2, scale=scale, split_k=sk) if sk > 1 else \
            hip.gemm(d1, d2, m, y2[r0:r1], d1, 0, g[r0:r1], d2, 0, G, d2, scale=scale, residual=G, ldr=d2)

    def run(chunks):
        step = -(-M // chunks)
        step = (step + 31) // 32 * 32
        for r0 in range(0, M, step):
            pair(r0, min(M, r0 + step))

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        return best

    only_d = timed(lambda: hip.gemm(M, d1, d2, g, d2, 1, w3, d2, 1, dz2, d1, mask=y2, ldmask=d1))
    print(f"M{M} {d1}->{d2}: input gradient alone {only_d:.1f} us; g is {M * d2 * 2 / 1e6:.0f} MB")
    for chunks in (() if FUSED_ONLY else (1, 2, 4, 8, 16)):
        print(f"   pair in {chunks:2d} chunk(s): {timed(lambda: run(chunks)):.1f} us")
    if (d1, d2) == (64, 256):
        scratch = torch.empty(hip.conv1x1_bwd_fused_scratch_floats(M), device=dev)
        print(f"   fused kernel (csrc/bwd_fused.hip): {timed(lambda: hip.conv1x1_bwd_fused(g, y2, w3, dz2, G, scratch, scale=scale, use_mask=True)):.1f} us"
              f" = {(M * (d2 + 2 * d1) * 2 + 2 * scratch.numel() * 4) / 1e6:.0f} MB")
