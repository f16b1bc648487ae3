"""Micro-benchmark of the bf16 GEMM kernels on the step's dominant shapes: every case is timed under a list of library tuning
settings (default: the product dispatch, and DETR_HIP_GEMM_STREAM=2 = tile engine only) with HIP events around `reps`
back-to-back launches, and the variants' results are compared with the first one.
usage: python scripts/micro_gemm.py out.json [--cold] [NAME=VAL[,NAME=VAL] ...]      (extra settings to time, e.g. DETR_HIP_GEMM_TILE=1)
--cold: every launch is timed on its own after a 600 MB flush of L2 / Infinity Cache (back-to-back launches of one shape run out
of the 256 MB Infinity Cache, which the training step's operands do not)
(The round-3 LDS-DMA experiment -- profiles/r03_micro_gemm_dma_experiment.txt -- was produced with an earlier form of this script.)"""
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
COLD = "--cold" in sys.argv
if COLD:
    sys.argv.remove("--cold")
hip.ensure_workspace(dev)
torch.manual_seed(0)
bf = torch.bfloat16


def operands(M, N, K, ak, bk):
    A = (torch.randn(M, K, device=dev) if ak else torch.randn(K, M, device=dev)).to(bf)
    B = ((torch.randn(N, K, device=dev) if bk else torch.randn(K, N, device=dev)) / K ** 0.5).to(bf)
    return A, B


def run(case, mode, reps=30):
    name, M, N, K, ak, bk, kw = case
    A, B = case_ops[name]
    torch.manual_seed(M + N + K)              # the same epilogue operands under every mode
    cdt = torch.float32 if kw.get("c32") else bf
    C = torch.zeros(M, N, device=dev, dtype=cdt)
    res = torch.randn(M, N, device=dev).to(torch.float32 if kw.get("r32") else bf) if kw.get("res") else None
    msk = torch.randn(M, N, device=dev).to(bf) if kw.get("mask") else None
    bias = torch.randn(N, device=dev) if kw.get("bias") else None
    sk = hip.pick_split_k(M, N, K) if kw.get("wgrad") else 1
    args = (M, N, K, A, A.stride(0), ak, B, B.stride(0), bk, C, N)
    kws = dict(bias=bias, residual=res, ldr=N if res is not None else 0, mask=msk, ldmask=N if msk is not None else 0,
               act=kw.get("act", 0), alpha=kw.get("alpha", 1.0), compute=1, split_k=sk)
    if kw.get("wgrad") and sk == 1:
        kws.update(residual=C, ldr=N)
    for k, v in mode.items():
        hip.set_tuning(k, v)
    try:
        for _ in range(3):
            if kw.get("wgrad"):
                C.zero_()
            hip.gemm(*args, **kws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        if COLD:
            for _ in range(7):
                FLUSH.zero_()
                e0.record()
                hip.gemm(*args, **kws)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = [sorted(ts)[len(ts) // 2]]
        else:
            for _ in range(3):
                e0.record()
                for _ in range(reps):
                    hip.gemm(*args, **kws)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / reps * 1e3)
        if kw.get("wgrad"):
            C.zero_()
            hip.gemm(*args, **kws)
            torch.cuda.synchronize()
        return min(ts), C.float().clone(), sk
    finally:
        for k in mode:
            hip.set_tuning(k, None)


hip.COMPUTE_BF16 = 1
FLUSH = torch.empty(600 * 1024 * 1024 // 4, device=dev) if COLD else None
cases = [
    ("l3 conv3 dgrad", 33600, 256, 1024, 1, 1, dict(mask=1)),
    ("l3 conv1 fwd", 33600, 256, 1024, 1, 0, dict(bias=1, act=1)),
    ("l2 conv3 dgrad", 133600, 128, 512, 1, 1, dict(mask=1)),
    ("l2 conv1 fwd", 133600, 128, 512, 1, 0, dict(bias=1, act=1)),
    ("l4 conv3 fwd", 8400, 2048, 512, 1, 0, dict(bias=1, res=1, act=1)),
    ("l4 conv3 dgrad", 8400, 512, 2048, 1, 1, dict(mask=1)),
    ("l3 down fwd", 33600, 1024, 512, 1, 0, dict(bias=1)),
    ("l3 conv1 dgrad b0", 33600, 1024, 512, 1, 1, dict(res=1, mask=1)),
    ("ffn lin2 fwd", 8400, 256, 2048, 1, 1, dict(bias=1, res=1, r32=1, c32=1)),
    ("ffn lin1 dgrad", 8400, 256, 2048, 1, 0, dict(res=1, r32=1, c32=1)),
    ("l3 conv3 wgrad", 256, 1024, 33600, 0, 0, dict(wgrad=1, c32=1)),
    ("l3 conv1 wgrad", 1024, 256, 33600, 0, 0, dict(wgrad=1, c32=1)),
    ("l2 conv3 wgrad", 128, 512, 133600, 0, 0, dict(wgrad=1, c32=1)),
    ("l2 conv1 wgrad", 512, 128, 133600, 0, 0, dict(wgrad=1, c32=1)),
    ("l1 conv3 wgrad", 64, 256, 534400, 0, 0, dict(wgrad=1, c32=1)),
    ("l1 conv1 wgrad", 256, 64, 534400, 0, 0, dict(wgrad=1, c32=1)),
    ("l3 down wgrad", 1024, 512, 33600, 0, 0, dict(wgrad=1, c32=1)),
    ("l4 conv3 wgrad", 512, 2048, 8400, 0, 0, dict(wgrad=1, c32=1)),
    ("ffn lin2 wgrad", 256, 2048, 8400, 0, 0, dict(wgrad=1, c32=1)),
    ("dec ffn lin2 fwd", 800, 256, 2048, 1, 1, dict(bias=1, res=1, r32=1, c32=1)),
    ("dec ffn lin1 dgrad", 800, 256, 2048, 1, 0, dict(res=1, r32=1, c32=1)),
    ("enc proj wgrad", 256, 256, 8400, 0, 0, dict(wgrad=1, c32=1)),
    ("ffn lin1 wgrad", 2048, 256, 8400, 0, 0, dict(wgrad=1, c32=1)),
    ("l3 conv3 fwd K256", 33600, 1024, 256, 1, 0, dict(bias=1, res=1, act=1)),
    ("l3 conv1 dgrad K256", 33600, 1024, 256, 1, 1, dict(res=1, mask=1)),
    ("l3 conv1 fwd b0 K512", 33600, 256, 512, 1, 0, dict(bias=1, act=1)),
    ("l2 conv3 fwd K128", 133600, 512, 128, 1, 0, dict(bias=1, res=1, act=1)),
    ("l2 conv1 dgrad K128", 133600, 512, 128, 1, 1, dict(res=1, mask=1)),
    ("l2 conv1 fwd K256", 133600, 128, 256, 1, 0, dict(bias=1, act=1)),
    ("l2 conv3 dgrad K256", 133600, 128, 256, 1, 1, dict(mask=1)),
    ("l1 conv3 fwd K64", 534400, 256, 64, 1, 0, dict(bias=1, res=1, act=1)),
    ("l1 conv1 fwd K256", 534400, 64, 256, 1, 0, dict(bias=1, act=1)),
    ("ffn lin1 fwd K256", 8400, 2048, 256, 1, 1, dict(bias=1, act=1)),
    ("ffn lin2 dgrad K256", 8400, 2048, 256, 1, 0, dict(mask=1, alpha=1.0 / 0.9)),
]
case_ops = {c[0]: operands(c[1], c[2], c[3], c[4], c[5]) for c in cases}
modes = [("default", {}), ("tile-engine", {"DETR_HIP_GEMM_STREAM": 2})]
for extra in sys.argv[2:]:
    modes.append((extra, {kv.split("=")[0]: kv.split("=")[1] for kv in extra.split(",")}))
rows = []
print(f"{'case':22s} {'shape':30s} {'sk':>4s} " + " ".join(f"{n[:18]:>18s}" for n, _ in modes) + "   (us; rel. max diff vs the first)")
for case in cases:
    name, M, N, K, ak, bk, kw = case
    res = [run(case, m) for _, m in modes]
    scale = float(res[0][1].abs().max()) + 1e-30
    cells = [f"{t:9.1f} {float((c - res[0][1]).abs().max()) / scale:8.1e}" for t, c, _ in res]
    print(f"{name:22s} M{M} N{N} K{K} a{ak} b{bk:<3d} {res[0][2]:4d} " + " ".join(cells))
    rows.append(dict(case=name, M=M, N=N, K=K, ak=ak, bk=bk, split_k=res[0][2], us={n: r[0] for (n, _), r in zip(modes, res)}))
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
