"""Timeline of the LDS-DMA halo convolution (csrc/conv_halo_dma.h) from a -DDETR_ABLATE=64 build: wave 0 of three workgroups (first, middle,
last of the grid) stamps s_memtime in front of the counted vmcnt wait (a), in front of the barrier (b) and behind it (c) in every step.
Prints per workgroup: prologue / loop / epilogue cycles, and per step the three gaps  c(s-1)->a(s) work | a->b DMA wait | b->c barrier wait.
usage: DETR_HIP_LIB=.../libdetr_hip_a64.so python scripts/experiments/conv_trace.py [C] [mode]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
import torch

from detr_tf import _hip as hip

lib = hip.load()
dev = "cuda"
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, H, W = {128: (8, 100, 167), 256: (8, 50, 84), 512: (8, 25, 42)}[C]
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
x, w, y = bf(N, H, W, C), bf(3, 3, C, C), torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    hip.conv3x3(mode, x, w, y, N, H, W, C, H, W, C, 1, compute=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    hip.conv3x3(mode, x, w, y, N, H, W, C, H, W, C, 1, compute=1)
e1.record()
torch.cuda.synchronize()
print(f"C={C} mode={mode}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch (events, 20 back to back)")
STEPS, NTR = 40, 40 * 3 + 6
out = (ctypes.c_longlong * (3 * NTR))()
lib.detr_hip_debug_conv_trace.argtypes = [ctypes.c_void_p]
assert lib.detr_hip_debug_conv_trace(out) == 0
nsteps = (C // 32) * 9
for g in range(3):
    t = list(out[g * NTR:(g + 1) * NTR])
    k0, l0, l1, k1 = t[STEPS * 3:STEPS * 3 + 4]
    r0, r1 = t[STEPS * 3 + 4:STEPS * 3 + 6]
    print(f"workgroup {g}: {(r1 - r0) * 10} ns by the 100 MHz counter -> core clock {(k1 - k0) / max(1, (r1 - r0) * 10):.2f} GHz")
    print(f"workgroup {g}: prologue {l0 - k0}  loop {l1 - l0}  epilogue {k1 - l1}  total {k1 - k0} ticks; loop / step {(l1 - l0) / nsteps:.1f}")
    rows = []
    for s in range(min(nsteps, STEPS)):
        a, b, c = t[3 * s:3 * s + 3]
        prev_c = t[3 * (s - 1) + 2] if s else l0
        rows.append((a - prev_c, b - a, c - b))
    n = len(rows)
    print("   mean over steps: work %.1f | dma wait %.1f | barrier wait %.1f" % tuple(sum(r[i] for r in rows) / n for i in range(3)))
    print("   steps 0..35 (work|dma|bar): " + " ".join(f"{r[0]}|{r[1]}|{r[2]}" for r in rows[:36]))
