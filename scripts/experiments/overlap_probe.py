"""Do the latency-bound attention backward kernels and the chip-filling weight-gradient GEMMs of a transformer layer overlap
when they are issued on two streams?  Times each alone, back to back on one stream, and concurrently on two."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
hip.ensure_workspace(dev)
hip.COMPUTE_BF16 = 1
B, H, T, D = 8, 8, 1050, 256
R = B * T
q, k, v, do = (torch.randn(R, D, device=dev) for _ in range(4))
o, lse = torch.zeros(R, D, device=dev), torch.zeros(B * H, T, device=dev)
hip.attention(q, k, v, o, lse, B, H, T, T, compute=1, scale=32 ** -0.5)
dq, dk, dv, delta = torch.zeros(R, D, device=dev), torch.zeros(R, D, device=dev), torch.zeros(R, D, device=dev), torch.zeros(B * H, T, device=dev)
x16, dh16 = torch.randn(R, D, device=dev).to(torch.bfloat16), torch.randn(R, 2048, device=dev).to(torch.bfloat16)
h16, dy = torch.randn(R, 2048, device=dev).to(torch.bfloat16), torch.randn(R, D, device=dev).to(torch.bfloat16)
gW1, gb1, gW2, gb2 = torch.zeros(2048, D, device=dev), torch.zeros(2048, device=dev), torch.zeros(D, 2048, device=dev), torch.zeros(D, device=dev)
ws2 = torch.empty(64 * 1024 * 1024, device=dev)


def attn_bwd():
    hip.attention(q, k, v, o, lse, B, H, T, T, compute=1, scale=32 ** -0.5, d_o=do, dq=dq, dk=dk, dv=dv, delta=delta)


def wgrads():
    hip.gemm_group([hip.linear_wgrad_call(dh16, x16, gW1, bias_grad=gb1), hip.linear_wgrad_call(dy, h16, gW2, bias_grad=gb2)])


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


side = torch.cuda.Stream()


def both_serial():
    attn_bwd()
    wgrads()


def both_concurrent():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        old = hip.WORKSPACE
        hip.WORKSPACE = ws2            # the side stream needs its own split-K slabs
        wgrads()
        hip.WORKSPACE = old
        done = torch.cuda.Event()
        done.record()
    attn_bwd()
    torch.cuda.current_stream().wait_event(done)


print(f"attention bwd alone {timeit(attn_bwd):.1f} us | FFN wgrad pair alone {timeit(wgrads):.1f} us | serial {timeit(both_serial):.1f} us | "
      f"two streams {timeit(both_concurrent):.1f} us", flush=True)
