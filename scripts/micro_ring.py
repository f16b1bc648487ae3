"""Micro-benchmark of the ring GEMM (csrc/gemm_ring.h) on the step's tall unsplit shapes: every case under a list of tuning settings
(HIP events around `reps` back-to-back launches, best of 3).  usage: python scripts/micro_ring.py out.json "NAME=V,NAME=V" ...
Each extra argument is one mode (comma-separated settings); "-" is the default dispatch.  Results are compared with the first mode."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    sys.path.insert(0, p)
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
hip.ensure_workspace(dev)
bf = torch.bfloat16
hip.COMPUTE_BF16 = 1
CASES = [
    ("l3 conv3 dgrad", 33600, 256, 1024, 1, dict(mask=1)),
    ("l3 conv1 fwd", 33600, 256, 1024, 0, dict(bias=1, act=1)),
    ("l2 conv3 dgrad", 133600, 128, 512, 1, dict(mask=1)),
    ("l2 conv1 fwd", 133600, 128, 512, 0, dict(bias=1, act=1)),
    ("l2 down fwd", 133600, 256, 512, 0, dict(bias=1)),
    ("l4 conv3 fwd", 8400, 2048, 512, 0, dict(bias=1, res=1, act=1)),
    ("l4 conv3 dgrad", 8400, 512, 2048, 1, dict(mask=1)),
    ("l3 down fwd", 33600, 1024, 512, 0, dict(bias=1)),
    ("l3 conv1 dgrad b0", 33600, 1024, 512, 1, dict(res=1, mask=1)),
    ("ffn lin2 fwd", 8400, 256, 2048, 1, dict(bias=1, res=1, r32=1, c32=1)),
    ("ffn lin1 dgrad", 8400, 256, 2048, 0, dict(res=1, r32=1, c32=1)),
    ("l3 conv1 fwd b0 K512", 33600, 256, 512, 0, dict(bias=1, act=1)),
]
if os.environ.get("RING_SMALL") == "1":          # the transformer's projection shapes (what bf16 twins of the LayerNorm outputs would put on the ring)
    CASES = [("enc qk proj", 8400, 512, 256, 1, dict(bias=1, c32=1)), ("enc v proj", 8400, 256, 256, 1, dict(bias=1, c32=1)),
             ("enc qkv dgrad", 8400, 256, 768, 0, dict(c32=1)), ("dec q proj", 800, 256, 256, 1, dict(bias=1, c32=1)),
             ("dec qk proj", 800, 512, 256, 1, dict(bias=1, c32=1)), ("dec kv proj all", 8400, 3072, 256, 1, dict(bias=1, c32=1)),
             ("dec ffn1", 800, 2048, 256, 1, dict(bias=1, act=1)), ("dec ffn2", 800, 256, 2048, 1, dict(bias=1, res=1, r32=1, c32=1)),
             ("dec out proj", 800, 256, 256, 1, dict(bias=1, res=1, r32=1, c32=1)), ("enc out proj", 8400, 256, 256, 1, dict(bias=1, res=1, r32=1, c32=1))]
only = os.environ.get("RING_CASES")
if only:
    CASES = [c for c in CASES if any(o in c[0] for o in only.split(","))]


F32 = os.environ.get("RING_F32") == "1"          # the fp32 parity mode: fp32 operands and tensors, compute = 0


def run(case, mode, reps=30):
    name, M, N, K, bk, kw = case
    if F32:
        global bf
        bf = torch.float32
        reps = 10
    torch.manual_seed(M + N + K + bk)
    A = torch.randn(M, K, device=dev).to(bf)
    B = ((torch.randn(N, K, device=dev) if bk else torch.randn(K, N, device=dev)) / K ** 0.5).to(bf)
    cdt = torch.float32 if kw.get("c32") else bf
    C = torch.zeros(M, N, device=dev, dtype=cdt)
    res = torch.randn(M, N, device=dev).to(torch.float32 if kw.get("r32") else bf) if kw.get("res") else None
    msk = torch.randn(M, N, device=dev).to(bf) if kw.get("mask") else None
    bias = torch.randn(N, device=dev) if kw.get("bias") else None
    args = (M, N, K, A, K, 1, B, B.stride(0), bk, C, N)
    kws = dict(bias=bias, residual=res, ldr=N if res is not None else 0, mask=msk, ldmask=N if msk is not None else 0, act=kw.get("act", 0),
               compute=0 if F32 else 1)
    for k, v in mode.items():
        hip.set_tuning(k, v)
    try:
        for _ in range(3):
            hip.gemm(*args, **kws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                hip.gemm(*args, **kws)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps * 1e3)
        return min(ts), C.float().clone()
    finally:
        for k in mode:
            hip.set_tuning(k, None)


modes = []
for extra in sys.argv[2:]:
    modes.append((extra, {} if extra == "-" else {kv.split("=")[0]: kv.split("=")[1] for kv in extra.split(",")}))
short = lambda n: n.replace("DETR_HIP_", "").replace("GEMM_", "")[:16]
print(f"{'case':22s} {'shape':26s} " + " ".join(f"{short(n):>16s}" for n, _ in modes) + "   (us, * = differs from the first)")
rows = []
plan = (hip.ctypes.c_int32 * 8)() if hasattr(hip, "ctypes") else None
for case in CASES:
    name, M, N, K, bk, kw = case
    res = [run(case, m) for _, m in modes]
    cells = [f"{t:15.1f}{' ' if torch.equal(c, res[0][1]) else '*'}" for t, c in res]
    print(f"{name:22s} M{M} N{N} K{K} b{bk:<2d} " + " ".join(cells))
    rows.append(dict(case=name, M=M, N=N, K=K, bk=bk, us={n: r[0] for (n, _), r in zip(modes, res)}))
json.dump(rows, open(sys.argv[1], "w"), indent=1)
