"""Micro-benchmark of the bf16 attention kernels on the step's shapes under a list of DETR_HIP_ATTN_SPLIT settings.
usage: python scripts/micro_attn.py [split ...]       (default: 0 = heuristic, 1, 2, 4)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    sys.path.insert(0, p)
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
splits = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4]
H, D = 8, 256
step = torch.tensor([0x1234567, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
print(f"{'shape':24s} " + " ".join(f"split{s:<3d} fwd   bwd  |" for s in splits) + "  (us, dropout 0.1)")
for B, T, S in ((8, 1050, 1050), (8, 100, 1050), (8, 100, 100), (16, 1050, 1050), (16, 300, 1050), (8, 1344, 1344)):
    torch.manual_seed(0)
    q, k, v, do = (torch.randn(B * n, D, device=dev) for n in (T, S, S, T))
    o, lse, delta = torch.zeros(B * T, D, device=dev), torch.zeros(B * H, T, device=dev), torch.zeros(B * H, T, device=dev)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    cells = []
    for sp in splits:
        hip.set_tuning("DETR_HIP_ATTN_SPLIT", sp if sp else None)
        fwd = lambda: hip.attention(q, k, v, o, lse, B, H, T, S, scale=32 ** -0.5, compute=1, dropout_p=0.1, dropout_site=3, dropout_step=step)
        bwd = lambda: hip.attention(q, k, v, o, lse, B, H, T, S, scale=32 ** -0.5, compute=1, dropout_p=0.1, dropout_site=3, dropout_step=step,
                                    d_o=do, dq=dq, dk=dk, dv=dv, delta=delta)
        ts = []
        for fn in (fwd, bwd):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            ts.append(best)
        cells.append(f"{ts[0]:8.1f} {ts[1]:6.1f}  |")
    hip.set_tuning("DETR_HIP_ATTN_SPLIT", None)
    print(f"B{B} T{T} S{S}".ljust(24) + " " + " ".join(cells))
