"""Timeline of the ring GEMM (csrc/gemm_ring.h) from a -DDETR_ABLATE=64 build of gemm_ring.hip: wave 0 of three workgroups (first, middle, last
of the grid) stamps s_memtime in front of the counted vmcnt wait (a), in front of the barrier (b) and behind it (c) in every K stage.
Prints per workgroup: prologue / loop / epilogue ticks, the core clock (s_memtime against the 100 MHz counter) and per stage the three gaps
c(s-1)->a(s) work | a->b request wait | b->c barrier wait.
usage: DETR_HIP_LIB=.../libdetr_hip_a64.so python scripts/experiments/ring_trace.py M N K [bk]"""
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
M, N, K = (int(v) for v in sys.argv[1:4])
bk = int(sys.argv[4]) if len(sys.argv) > 4 else 1
bf = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(torch.bfloat16)
A, B, C = bf(M, K), (bf(N, K) if bk else bf(K, N)), torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
run = lambda: hip.gemm(M, N, K, A, K, 1, B, K if bk else N, bk, C, N, compute=1)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
plan = (ctypes.c_int32 * 8)()
lib.detr_hip_gemm_ring_plan(M, N, K, plan)
print(f"M{M} N{N} K{K} bk={bk}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch; plan tm,tn,ns,rows,tiles_m,tiles_n,wgs,lds = {list(plan)}")
STEPS, NTR = 40, 40 * 3 + 6
out = (ctypes.c_longlong * (3 * NTR))()
lib.detr_hip_debug_ring_trace.argtypes = [ctypes.c_void_p]
assert lib.detr_hip_debug_ring_trace(out) == 0
nst = K // 64
for g in range(3):
    t = list(out[g * NTR:(g + 1) * NTR])
    k0, l0, l1, k1, r0, r1 = t[STEPS * 3:STEPS * 3 + 6]
    print(f"workgroup {g}: {(r1 - r0) * 10} ns -> core clock {(k1 - k0) / max(1, (r1 - r0) * 10):.2f} GHz; prologue {l0 - k0}  loop {l1 - l0}  epilogue {k1 - l1}  total {k1 - k0} ticks; loop / stage {(l1 - l0) / nst:.1f}")
    rows = []
    for s in range(min(nst, STEPS)):
        a, b, c = t[3 * s:3 * s + 3]
        prev_c = t[3 * (s - 1) + 2] if s else l0
        rows.append((a - prev_c, b - a, c - b))
    n = len(rows)
    print("   mean over stages: work %.1f | request wait %.1f | barrier wait %.1f" % tuple(sum(r[i] for r in rows) / n for i in range(3)))
    print("   stages (work|wait|bar): " + " ".join(f"{r[0]}|{r[1]}|{r[2]}" for r in rows[:32]))
