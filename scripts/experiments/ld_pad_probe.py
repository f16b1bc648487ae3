"""Does the leading dimension matter?  Times bf16 tile-GEMM launches of the step with packed operands (row pitch = a power of two:
every row of a tile lands on the same L2 / HBM channel if the channel interleave is not hashed) against the same operands
with a padded row pitch (+64 elements).  Cold: a 600 MB flush before every timed launch.  python scripts/experiments/ld_pad_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    sys.path.insert(0, p)
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
hip.ensure_workspace(dev)
bf = torch.bfloat16
FLUSH = torch.empty(600 * 1024 * 1024 // 4, device=dev)


def padded(rows, cols, pad):
    t = torch.randn(rows, cols + pad, device=dev).to(bf)
    return t[:, :cols]


def time_one(fn, cold=True, n=9):
    ts = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(n):
        if cold:
            FLUSH.zero_()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


CASES = [  # name, M, N, K, ak, bk, res, mask, wgrad
    ("enc ffn2 fwd", 8400, 256, 2048, 1, 1, True, False, False),
    ("layer3 conv3 dgrad", 33600, 256, 1024, 1, 1, False, True, False),
    ("layer2 conv3 dgrad", 133600, 128, 512, 1, 1, False, True, False),
    ("layer3 conv1 wgrad", 256, 1024, 33600, 0, 0, False, False, True),
    ("dec ffn2 fwd", 800, 256, 2048, 1, 1, True, False, False),
]
for name, M, N, K, ak, bk, res, mask, wgrad in CASES:
    out = []
    for pad_a, pad_b, pad_c in ((0, 0, 0), (64, 0, 0), (64, 64, 0), (64, 64, 64)):
        A = padded(M, K, pad_a) if ak else padded(K, M, pad_a)
        B = padded(N, K, pad_b) if bk else padded(K, N, pad_b)
        cdt = torch.float32 if wgrad else bf
        Cfull = torch.zeros(M, N + pad_c, device=dev, dtype=cdt)
        C = Cfull[:, :N]
        R = padded(M, N, pad_c) if res else None
        Mk = padded(M, N, pad_c) if mask else None
        sk = hip.pick_split_k(M, N, K) if wgrad else 1
        hip.COMPUTE_BF16 = 1
        kws = dict(residual=R, ldr=R.stride(0) if R is not None else 0, mask=Mk, ldmask=Mk.stride(0) if Mk is not None else 0,
                   compute=1, split_k=sk)
        fn = lambda: hip.gemm(M, N, K, A, A.stride(0), ak, B, B.stride(0), bk, C, C.stride(0), **kws)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        out.append((pad_a, pad_b, pad_c, time_one(fn, cold=True), time_one(fn, cold=False)))
    print(f"{name:22s} M{M} N{N} K{K} ak{ak} bk{bk}  " + "  ".join(f"pad({a},{b},{c}): cold {tc:6.1f} warm {tw:6.1f} us" for a, b, c, tc, tw in out), flush=True)
