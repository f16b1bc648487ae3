"""Calibration probe: what does a plain elementwise kernel achieve on this GPU for the traffic mixes of the
short-K 1x1-conv GEMMs (read A + residual, write C)?  torch elementwise kernels on bf16 [534400, 256]."""
import torch
dev = "cuda"
M, N = 534400, 256
a = torch.randn(M, N, device=dev).to(torch.bfloat16)
b = torch.randn(M, N, device=dev).to(torch.bfloat16)
c = torch.empty_like(a)
s = torch.randn(M, 64, device=dev).to(torch.bfloat16)


def t(fn, reps=20):
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


nb = a.numel() * 2
for name, fn, byts in [("copy  (1R 1W)", lambda: c.copy_(a), 2 * nb), ("add   (2R 1W)", lambda: torch.add(a, b, out=c), 3 * nb),
                       ("relu_ (1R 1W inplace)", lambda: c.relu_(), 2 * nb), ("fill  (0R 1W)", lambda: c.zero_(), nb),
                       ("sum   (1R 0W)", lambda: a.sum(), nb),
                       ("addcmul (3R 1W)", lambda: torch.addcmul(a, b, c, out=c), 4 * nb)]:
    us = t(fn)
    print(f"{name:24s} {us:8.1f} us  {byts / us / 1e6:6.2f} TB/s")
af, bf, cf = a.float(), b.float(), c.float()
nb4 = af.numel() * 4
for name, fn, byts in [("copy f32", lambda: cf.copy_(af), 2 * nb4), ("add f32", lambda: torch.add(af, bf, out=cf), 3 * nb4)]:
    us = t(fn)
    print(f"{name:24s} {us:8.1f} us  {byts / us / 1e6:6.2f} TB/s")
