"""GPU-side tuning sweep: times representative GEMM / conv3x3 shapes of the DETR step under every
tile configuration (DETR_HIP_GEMM_TILE / DETR_HIP_CONV_TILE hooks) and split-K count.
Output: one line per (shape, config) sorted per shape; used to set the dispatch heuristics."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "detr-tensorflow_amd"))
import torch

from detr_tf import _hip as hip

hip.load()
dev = "cuda"
ws = hip.ensure_workspace(dev)
TILES = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x64", 4: "128x32", 5: "64x256|64x128(bf16)", 6: "256x64"}


def timeit(fn, reps=8, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def gemm_case(M, N, K, ak, bk, batch=1, res=False, splits=(1,), tiles=(0, 1, 2, 3)):
    A = torch.randn((M, K) if ak else (K, M), device=dev)
    Bm = torch.randn((N, K) if bk else (K, N), device=dev)
    if batch > 1:   # attention layout is emulated with independent dense batches
        A = torch.randn((batch,) + tuple(A.shape), device=dev)
        Bm = torch.randn((batch,) + tuple(Bm.shape), device=dev)
    C = torch.zeros((batch, M, N) if batch > 1 else (M, N), device=dev)
    R = torch.randn(M, N, device=dev) if res else None
    if os.environ.get("TUNE_ACT16") == "1" and batch == 1:      # bf16 storage: activations (A, C, residual) and weights (B)
        A, C = A.to(torch.bfloat16), C.to(torch.bfloat16)
        R = R.to(torch.bfloat16) if R is not None else None
        if len(splits) == 1 and splits[0] == 1:
            Bm = Bm.to(torch.bfloat16)
        else:
            Bm, C = Bm.to(torch.bfloat16), C.float()          # wgrad form: both operands activations, fp32 output
    lda, ldb = (K if ak else M), (K if bk else N)
    out = []
    for sk in splits:
        for t in tiles:
            os.environ["DETR_HIP_GEMM_TILE"] = str(t)
            kw = dict(split_k=sk)
            if batch > 1:
                kw.update(batch=batch, batch_inner=1, sA=(A[0].numel(), 0), sB=(Bm[0].numel(), 0), sC=(M * N, 0))
            if res and sk == 1:
                kw.update(residual=R, ldr=N)
            try:
                ms = timeit(lambda: hip.gemm(M, N, K, A, lda, ak, Bm, ldb, bk, C, N, **kw))
            except RuntimeError as e:
                ms = float("nan")
            out.append((ms, sk, TILES[t]))
    os.environ["DETR_HIP_GEMM_TILE"] = "0"
    fl = 2.0 * M * N * K * batch
    out.sort()
    tag = f"gemm M{M} N{N} K{K} b{batch} ak{ak} bk{bk}{' res' if res else ''}"
    print(tag + " :: " + " | ".join(f"{ms*1e3:7.1f}us sk{sk} {t} {fl/ms/1e9:5.1f}TF" for ms, sk, t in out[:8]), flush=True)


def conv_case(N, H, W, Ci, Co, stride, mode, tiles=(0, 1, 2, 3)):
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = torch.randn(N, H, W, Ci, device=dev)
    w = torch.randn(3, 3, Ci, Co, device=dev)
    y = torch.zeros(N, Ho, Wo, Co, device=dev)
    out = []
    for t in tiles:
        os.environ["DETR_HIP_CONV_TILE"] = str(t)
        if mode == 0:
            ms = timeit(lambda: hip.conv3x3(0, x, w, y, N, H, W, Ci, Ho, Wo, Co, stride, act=1))
        else:
            ms = timeit(lambda: hip.conv3x3(1, y, w, x, N, H, W, Ci, Ho, Wo, Co, stride))
        out.append((ms, TILES[t]))
    os.environ["DETR_HIP_CONV_TILE"] = "0"
    rows = N * (H * W if mode == 1 else Ho * Wo)
    fl = 2.0 * rows * 9 * Ci * Co
    out.sort()
    print(f"conv mode{mode} N{N} {H}x{W}x{Ci}->{Co} s{stride} :: " + " | ".join(f"{ms*1e3:7.1f}us {t} {fl/ms/1e9:5.1f}TF" for ms, t in out),
          flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fwd", "wgrad", "attn", "conv"]
    if "bf16" in which:       # bf16-compute kernels: tiles 1 (=128x128) and 3 (=64x64) exist
        hip.COMPUTE_BF16 = 1
        which = [w for w in which if w != "bf16"] or ["fwd", "wgrad", "conv"]
        for (M, N, K, ak, bk, res) in [(33600, 256, 1024, 1, 0, False), (33600, 1024, 256, 1, 0, True), (33600, 256, 1024, 1, 1, False),
                                       (133600, 512, 128, 1, 0, True), (133600, 128, 512, 1, 0, False), (534400, 256, 64, 1, 0, True),
                                       (534400, 64, 256, 1, 0, False), (8400, 256, 256, 1, 1, False), (8400, 2048, 256, 1, 1, False),
                                       (8400, 256, 2048, 1, 1, True), (800, 256, 2048, 1, 1, True)]:
            gemm_case(M, N, K, ak, bk, res=res, tiles=(1, 2, 3, 5))
        for (M, N, K, sks) in [(256, 256, 800, (1, 3, 6, 12)), (256, 256, 8400, (8, 16, 32, 64)), (256, 1024, 33600, (8, 16, 32)), (64, 256, 534400, (64, 128, 256)),
                               (128, 512, 133600, (32, 64, 128)), (147, 64, 2134400, (128, 341, 682)), (2048, 256, 8400, (4, 8, 16))]:
            gemm_case(M, N, K, 0, 0, splits=sks, tiles=(1, 3))
        for mode in (0, 1):
            for shp in [(8, 50, 84, 256, 256, 1), (8, 25, 42, 512, 512, 1), (8, 100, 167, 128, 128, 1), (8, 200, 334, 64, 64, 1)]:
                conv_case(*shp, mode, tiles=(1, 3))
        sys.exit(0)
    if "fwd" in which:
        gemm_case(33600, 256, 1024, 1, 0)
        gemm_case(33600, 1024, 256, 1, 0, res=True)
        gemm_case(33600, 256, 1024, 1, 1)
        gemm_case(133600, 512, 128, 1, 0, res=True)
        gemm_case(133600, 128, 512, 1, 0)
        gemm_case(534400, 256, 64, 1, 0, res=True)
        gemm_case(534400, 64, 256, 1, 0, tiles=(0, 2, 3))
        gemm_case(8400, 256, 256, 1, 1)
        gemm_case(8400, 2048, 256, 1, 1)
        gemm_case(8400, 256, 2048, 1, 1, res=True)
        gemm_case(8400, 512, 2048, 1, 1)
        gemm_case(800, 256, 256, 1, 1, splits=(1,), tiles=(0, 2, 3))
        gemm_case(800, 256, 2048, 1, 1, res=True, tiles=(0, 2, 3))
    if "wgrad" in which:
        gemm_case(256, 256, 8400, 0, 0, splits=(8, 16, 32, 65), tiles=(1, 2, 3))
        gemm_case(256, 256, 800, 0, 0, splits=(1, 3, 6, 12), tiles=(1, 3))
        gemm_case(256, 1024, 33600, 0, 0, splits=(16, 32, 64), tiles=(1, 2, 3))
        gemm_case(2048, 256, 8400, 0, 0, splits=(8, 16, 32), tiles=(1, 2, 3))
        gemm_case(64, 256, 534400, 0, 0, splits=(256, 512, 1024), tiles=(3, 5))
        gemm_case(256, 64, 534400, 0, 0, splits=(256, 512, 1024), tiles=(3, 6))
        gemm_case(128, 512, 133600, 0, 0, splits=(128, 256, 512), tiles=(1, 3))
        gemm_case(147, 64, 2134400, 0, 0, splits=(341, 682, 1024, 2048), tiles=(3, 6))
    if "attn" in which:
        gemm_case(1050, 1050, 32, 1, 1, batch=64, tiles=(0, 1, 2, 3))
        gemm_case(1050, 32, 1050, 1, 0, batch=64, tiles=(0, 4))
        gemm_case(1050, 32, 1050, 0, 0, batch=64, tiles=(0, 4))
    if "conv" in which:
        for mode in (0, 1):
            conv_case(8, 50, 84, 256, 256, 1, mode)
            conv_case(8, 25, 42, 512, 512, 1, mode)
            conv_case(8, 100, 167, 128, 128, 1, mode)
            conv_case(8, 200, 334, 64, 64, 1, mode, tiles=(0, 2, 3))
