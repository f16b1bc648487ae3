"""Micro-benchmark of the bf16 3x3 convolutions on the step's shapes (forward with bias + ReLU, input gradient with the ReLU
mask, weight gradient): HIP events around `reps` back-to-back launches, or --cold: each launch alone after a 600 MB flush.
usage: python scripts/micro_conv.py [--cold] [NAME=VAL[,NAME=VAL] ...]      (extra library tuning settings to time)
DETR_HIP_LIB=<path> selects an experimental build of the library (e.g. one compiled with -DDETR_ABLATE=<bits>)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    sys.path.insert(0, p)
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
COLD = "--cold" in sys.argv
if COLD:
    sys.argv.remove("--cold")
hip.ensure_workspace(dev)
hip.COMPUTE_BF16 = 1
bf = torch.bfloat16
FLUSH = torch.empty(600 * 1024 * 1024 // 4, device=dev) if COLD else None
shapes = [(8, 200, 334, 64, 64, 1), (8, 100, 167, 128, 128, 1), (8, 50, 84, 256, 256, 1), (8, 25, 42, 512, 512, 1),
          (8, 200, 334, 128, 128, 2), (8, 100, 167, 256, 256, 2), (8, 50, 84, 512, 512, 2)]
modes = [("default", {})]
for extra in sys.argv[1:]:
    modes.append((extra, {kv.split("=")[0]: kv.split("=")[1] for kv in extra.split(",")}))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    if COLD:
        for _ in range(7):
            FLUSH.zero_()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return min(ts)


print(f"{'shape':34s} {'mode':6s} " + " ".join(f"{n[:22]:>22s}" for n, _ in modes) + "   (us  TFLOP/s)")
for N, H, W, Ci, Co, st in shapes:
    Ho, Wo = (H + 2 - 3) // st + 1, (W + 2 - 3) // st + 1
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Ci, device=dev).to(bf)
    w = (torch.randn(3, 3, Ci, Co, device=dev) / (3 * Ci ** 0.5)).to(bf)
    y = torch.zeros(N, Ho, Wo, Co, device=dev, dtype=bf)
    dy = torch.randn(N, Ho, Wo, Co, device=dev).to(bf)
    dx = torch.zeros(N, H, W, Ci, device=dev, dtype=bf)
    dw = torch.zeros(3, 3, Ci, Co, device=dev)
    bias = torch.randn(Co, device=dev)
    flops = 2.0 * N * Ho * Wo * 9 * Ci * Co
    calls = (("fwd", lambda: hip.conv3x3(0, x, w, y, N, H, W, Ci, Ho, Wo, Co, st, bias=bias, act=1, compute=1)),
             ("dgrad", lambda: hip.conv3x3(1, dy, w, dx, N, H, W, Ci, Ho, Wo, Co, st, mask=x, compute=1)),
             ("wgrad", lambda: hip.conv3x3(2, x, dy, dw, N, H, W, Ci, Ho, Wo, Co, st, compute=1)))
    for name, fn in calls:
        cells = []
        for _, m in modes:
            for k, v in m.items():
                hip.set_tuning(k, v)
            try:
                t = timed(fn)
            finally:
                for k in m:
                    hip.set_tuning(k, None)
            cells.append(f"{t:12.1f} {flops / t / 1e6:9.0f}")
        print(f"N{N} {H}x{W}x{Ci}->{Ho}x{Wo}x{Co} s{st}".ljust(34) + f" {name:6s} " + " ".join(cells))
