"""Runs the all-bf16 attention kernels (forward, backward, keep-bit generator) of one shape a few times: the workload of the
rocprofv3 passes in scripts/experiments/attn2_prof.sh.  usage: attn2_prof.py [B T S p reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    sys.path.insert(0, p)
import torch

from detr_tf import _hip as hip

hip.load()
B, T, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 1050, 1050)
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
H, D, dev = 8, 256, "cuda"
torch.manual_seed(0)
step = torch.tensor([0x1234567, 0, 0, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
q, k, v, do = (torch.randn(B * n, D, device=dev).to(torch.bfloat16) for n in (T, S, S, T))
o = torch.zeros(B * T, D, dtype=torch.bfloat16, device=dev)
lse, delta = torch.zeros(B * H, T, device=dev), torch.zeros(2 * B * H, T, device=dev)
dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
mask = torch.zeros(hip.attention_dropmask_words(B, H, T, S), dtype=torch.int32, device=dev) if p > 0 else None
kw = dict(scale=32 ** -0.5, dropout_p=p, dropout_site=3, dropout_step=step, dropmask=mask)
for _ in range(reps):
    if p > 0:
        hip.attention_dropmask(mask, B, H, T, S, dropout_p=p, dropout_site=3, dropout_step=step)
    hip.attention(q, k, v, o, lse, B, H, T, S, **kw)
    hip.attention(q, k, v, o, lse, B, H, T, S, d_o=do, dq=dq, dk=dk, dv=dv, delta=delta, **kw)
torch.cuda.synchronize()
