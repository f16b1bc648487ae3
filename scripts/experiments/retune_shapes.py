"""Re-times every GEMM shape of the bf16 step (profiles/r02_gemm_shapes.json, tile-engine families) under the forced tile
configurations (DETR_HIP_GEMM_TILE) and neighbouring split-K counts, and prints the shapes where the dispatch heuristic of
gemm_f32.hip / pick_split_k is more than 5 % off the best configuration."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
hip.ensure_workspace(dev)
TILES = {0: "auto", 3: "64x64", 1: "128x128", 5: "64x128", 2: "128x64"}


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


rows = json.load(open(os.path.join(ROOT, "profiles", "r02_gemm_shapes.json")))["rows"]
total_auto = total_best = 0.0
for r in rows:
    m = re.match(r"gemm_bf16c_kernel<(\w+), (\w+), (\w+), (\w+)>", r["family"])
    if not m:
        continue
    ak, bk, a16, b16 = (x == "true" for x in m.groups())
    s = re.match(r"(group\d+: )?M(\d+) N(\d+) K(\d+) b1 ak\d bk\d sk(\d+)( res)?( mask)?", r["shape"])
    if not s:
        continue
    M, N, K, sk = int(s.group(2)), int(s.group(3)), int(s.group(4)), int(s.group(5))
    res, mask = bool(s.group(6)), bool(s.group(7))
    dt = lambda f: torch.bfloat16 if f else torch.float32
    A = torch.randn((M, K) if ak else (K, M), device=dev).to(dt(a16))
    Bm = torch.randn((N, K) if bk else (K, N), device=dev).to(dt(b16))
    c16 = a16 and sk == 1                      # activations out in bf16 when the input activations are bf16 (forward / dgrad)
    C = torch.zeros(M, N, device=dev, dtype=dt(c16))
    R = torch.randn(M, N, device=dev).to(dt(c16)) if res else None
    Mk = torch.randn(M, N, device=dev).to(dt(c16)) if mask else None
    out = []
    for t in TILES:
        for skk in sorted({max(1, sk // 2), sk, sk * 2} if sk > 1 else {1}):
            os.environ["DETR_HIP_GEMM_TILE"] = str(t)
            kw = dict(split_k=skk, compute=1)
            if skk == 1:
                kw.update(residual=R, ldr=N if res else 0, mask=Mk, ldmask=N if mask else 0)
            try:
                us = timeit(lambda: hip.gemm(M, N, K, A, K if ak else M, int(ak), Bm, K if bk else N, int(bk), C, N, **kw))
            except RuntimeError:
                continue
            out.append((us, TILES[t], skk))
    os.environ["DETR_HIP_GEMM_TILE"] = "0"
    auto = [o for o in out if o[1] == "auto" and o[2] == sk][0][0]
    best = min(out)
    n = r["launches"]
    total_auto += auto * n
    total_best += best[0] * n
    flag = "  <<<" if best[0] < 0.95 * auto else ""
    print(f"{r['shape'][:58]:58s} x{n:2d} auto {auto:7.1f} us | best {best[0]:7.1f} us {best[1]} sk{best[2]}{flag}", flush=True)
print(f"sum over the step: auto {total_auto / 1e3:.3f} ms, best-per-shape {total_best / 1e3:.3f} ms")
