"""Micro-benchmark: the all-bf16 attention kernels (csrc/attention_dma.hip, io_dtype = 1) next to the round-3 kernels
(fp32 tensors, bf16 MFMA) on the step's shapes, under a list of DETR_HIP_ATTN_SPLIT settings for the new ones.
usage: python scripts/micro_attn2.py [split ...]       (default: 0 = heuristic)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    sys.path.insert(0, p)
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
splits = [int(x) for x in sys.argv[1:]] or [0]
H, D = 8, 256
step = torch.tensor([0x1234567, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)


def timeit(fn, inner=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / inner * 1e3)
    return best


print(f"{'shape':22s} {'p':>4s} | old fwd    bwd | mask | " + " | ".join(f"split{s}: fwd    bwd" for s in splits) + "   (us)")
SHAPES = ((8, 1050, 1050), (8, 100, 1050), (8, 100, 100), (16, 1050, 1050), (16, 300, 1050), (8, 1344, 1344))
for B, T, S in SHAPES[:int(os.environ.get('MICRO_SHAPES', '6'))]:
    for p in (0.1, 0.0):
        torch.manual_seed(0)
        q, k, v, do = (torch.randn(B * n, D, device=dev) for n in (T, S, S, T))
        o, lse, delta = torch.zeros(B * T, D, device=dev), torch.zeros(B * H, T, device=dev), torch.zeros(2 * B * H, T, device=dev)
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        kw = dict(scale=32 ** -0.5, dropout_p=p, dropout_site=3, dropout_step=step)
        old_f = timeit(lambda: hip.attention(q, k, v, o, lse, B, H, T, S, compute=1, **kw))
        old_b = timeit(lambda: hip.attention(q, k, v, o, lse, B, H, T, S, compute=1, d_o=do, dq=dq, dk=dk, dv=dv, delta=delta, **kw))
        q16, k16, v16, do16 = ((t * (32 ** -0.5 * 1.4426950408889634 if i == 0 else 1.0)).to(torch.bfloat16) for i, t in enumerate((q, k, v, do)))
        o16 = torch.zeros(B * T, D, dtype=torch.bfloat16, device=dev)
        dq16, dk16, dv16 = torch.zeros_like(q16), torch.zeros_like(k16), torch.zeros_like(v16)
        mask, t_mask = None, 0.0
        if p > 0.0:
            mask = torch.zeros(hip.attention_dropmask_words(B, H, T, S), dtype=torch.int32, device=dev)
            t_mask = timeit(lambda: hip.attention_dropmask(mask, B, H, T, S, dropout_p=p, dropout_site=3, dropout_step=step))
        cells = []
        for sp in splits:
            hip.set_tuning("DETR_HIP_ATTN_SPLIT", sp if sp else None)
            f = timeit(lambda: hip.attention(q16, k16, v16, o16, lse, B, H, T, S, dropmask=mask, **kw))
            b = timeit(lambda: hip.attention(q16, k16, v16, o16, lse, B, H, T, S, dropmask=mask, d_o=do16, dq=dq16, dk=dk16, dv=dv16, delta=delta, **kw))
            cells.append(f"{f:11.1f} {b:6.1f}")
        hip.set_tuning("DETR_HIP_ATTN_SPLIT", None)
        print(f"B{B} T{T} S{S}".ljust(22) + f" {p:4.1f} | {old_f:7.1f} {old_b:6.1f} | {t_mask:4.1f} | " + " | ".join(cells), flush=True)
