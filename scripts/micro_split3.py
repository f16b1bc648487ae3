"""f32x3 (detr_gemm_desc.compute = 2: fp32 accuracy on the bf16 matrix pipe, csrc/gemm_core.h mma_ktile_split3) against the exact fp32
MFMA kernels (compute = 0) on the fp32 step's GEMM and 3x3-convolution shapes: time per launch (HIP events around `reps` back-to-back
launches, best of 3) and error against an fp64 reference (max |err| / max |ref| and rms err / rms ref), bf16 MFMA (compute = 1) beside
them for scale.
usage: python scripts/micro_split3.py [out.json] [NAME=VAL ...]        (extra library tuning settings for the compute = 2 column)"""
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
extra = dict(a.split("=") for a in sys.argv[2:] if "=" in a)

# (name, M, N, K, a_kcontig, b_kcontig, wgrad)
GEMMS = [
    ("l1 c1 64<-256 M534400", 534400, 64, 256, 1, 0, 0),
    ("l1 c3 256<-64 M534400", 534400, 256, 64, 1, 0, 0),
    ("l2 c3 512<-128 M133600", 133600, 512, 128, 1, 0, 0),
    ("l3 c1 256<-1024 M33600", 33600, 256, 1024, 1, 0, 0),
    ("l3 c3 1024<-256 M33600", 33600, 1024, 256, 1, 0, 0),
    ("l4 c3 2048<-512 M8400", 8400, 2048, 512, 1, 0, 0),
    ("l4 c1 512<-2048 M8400", 8400, 512, 2048, 1, 0, 0),
    ("input_proj 256<-2048 M8400", 8400, 256, 2048, 1, 0, 0),
    ("enc qk 512<-256 M8400", 8400, 512, 256, 1, 1, 0),
    ("enc ffn1 2048<-256 M8400", 8400, 2048, 256, 1, 1, 0),
    ("enc ffn2 256<-2048 M8400", 8400, 256, 2048, 1, 1, 0),
    ("dec ffn1 2048<-256 M800", 800, 2048, 256, 1, 1, 0),
    ("dec proj 256<-256 M800", 800, 256, 256, 1, 1, 0),
    ("wgrad l3 256x1024 K33600", 256, 1024, 33600, 0, 0, 1),
    ("wgrad l1 64x256 K534400", 64, 256, 534400, 0, 0, 1),
    ("wgrad ffn 2048x256 K8400", 2048, 256, 8400, 0, 0, 1),
    ("wgrad l4 512x2048 K8400", 512, 2048, 8400, 0, 0, 1),
]
# (name, N, H, W, Ci, Co, stride)
CONVS = [
    ("l1 3x3 64ch 200x334", 8, 200, 334, 64, 64, 1),
    ("l2 3x3 128ch 100x167", 8, 100, 167, 128, 128, 1),
    ("l3 3x3 256ch 50x84", 8, 50, 84, 256, 256, 1),
    ("l4 3x3 512ch 25x42", 8, 25, 42, 512, 512, 1),
    ("l2 3x3 128ch s2 200x334", 8, 200, 334, 128, 128, 2),
]


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def errs(out, ref):
    d = (out.double() - ref)
    return float(d.abs().max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())


def set_extra(on):
    for k, v in extra.items():
        hip.set_tuning(k, int(v) if on else None)


rows = []
print(f"{'GEMM':34s} | {'fp32 us':>8s} {'f32x3 us':>8s} {'x':>5s} | max-rel / rms-rel error vs fp64:  exact fp32 | f32x3 | bf16")
for name, M, N, K, ak, bk, wg in GEMMS:
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev) if ak else torch.randn(K, M, device=dev)
    B = (torch.randn(N, K, device=dev) if bk else torch.randn(K, N, device=dev)) / K ** 0.5
    small = M * N <= (1 << 24) or K <= 2048
    ref = None
    if M * K <= (1 << 27):                  # fp64 reference where it is affordable
        Ad = (A if ak else A.t()).double()
        Bd = (B.t() if bk else B).double()
        ref = Ad @ Bd
    res = {}
    for mode in (0, 2, 1):
        C = torch.zeros(M, N, device=dev)
        hip.COMPUTE_BF16 = mode
        sk = hip.pick_split_k(M, N, K) if wg else 1
        kws = dict(compute=mode, split_k=sk)
        if wg and sk == 1:
            kws.update(residual=C, ldr=N)

        def fn():
            hip.gemm(M, N, K, A, A.stride(0), ak, B, B.stride(0), bk, C, N, **kws)
        set_extra(mode == 2)
        t = timed(fn, 10 if M * N * K > 1e11 else 30)
        C.zero_()
        fn()
        torch.cuda.synchronize()
        set_extra(False)
        res[mode] = (t,) + (errs(C, ref) if ref is not None else (float("nan"), float("nan")))
        del C
    hip.COMPUTE_BF16 = 0
    r0, r2, r1 = res[0], res[2], res[1]
    print(f"{name:34s} | {r0[0]:8.1f} {r2[0]:8.1f} {r0[0] / r2[0]:5.2f} | {r0[1]:.2e} / {r0[2]:.2e} | {r2[1]:.2e} / {r2[2]:.2e} | {r1[1]:.2e} / {r1[2]:.2e}", flush=True)
    rows.append(dict(kind="gemm", name=name, M=M, N=N, K=K, us_fp32=r0[0], us_f32x3=r2[0], us_bf16=r1[0], err_fp32=r0[1:], err_f32x3=r2[1:], err_bf16=r1[1:]))
    del A, B, ref

print(f"\n{'conv3x3 (fwd | dgrad | wgrad)':34s} | fp32 us -> f32x3 us per direction | rms-rel error vs fp64 (forward): exact | f32x3")
for name, Nb, H, W, Ci, Co, st in CONVS:
    torch.manual_seed(H + Ci)
    Ho, Wo = (H + 2 - 3) // st + 1, (W + 2 - 3) // st + 1
    x = torch.randn(Nb, H, W, Ci, device=dev)
    w = torch.randn(3, 3, Ci, Co, device=dev) / (9 * Ci) ** 0.5
    dy = torch.randn(Nb, Ho, Wo, Co, device=dev)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), stride=st, padding=1).permute(0, 2, 3, 1)
    out = {}
    for mode in (0, 2):
        hip.COMPUTE_BF16 = mode
        set_extra(mode == 2)
        y = torch.zeros(Nb, Ho, Wo, Co, device=dev)
        dx = torch.zeros_like(x)
        dw = torch.zeros_like(w)
        dims = (Nb, H, W, Ci, Ho, Wo, Co, st)
        t_f = timed(lambda: hip.conv3x3(0, x, w, y, *dims, compute=mode), 10)
        t_d = timed(lambda: hip.conv3x3(1, dy, w, dx, *dims, compute=mode), 10)
        t_w = timed(lambda: (dw.zero_(), hip.conv3x3(2, x, dy, dw, *dims, compute=mode)), 10)
        torch.cuda.synchronize()
        set_extra(False)
        out[mode] = (t_f, t_d, t_w, errs(y, ref)[1])
    hip.COMPUTE_BF16 = 0
    a, b = out[0], out[2]
    print(f"{name:34s} | {a[0]:7.1f} -> {b[0]:7.1f} | {a[1]:7.1f} -> {b[1]:7.1f} | {a[2]:7.1f} -> {b[2]:7.1f} | {a[3]:.2e} | {b[3]:.2e}", flush=True)
    rows.append(dict(kind="conv", name=name, us_fp32=a[:3], us_f32x3=b[:3], rms_fp32=a[3], rms_f32x3=b[3]))
if len(sys.argv) > 1 and "=" not in sys.argv[1]:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
