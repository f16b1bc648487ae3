"""Runs one f32x3 GEMM shape a few times: the workload of the rocprofv3 counter passes in scripts/experiments/x3_prof.sh.
usage: x3_prof.py [M N K ak bk reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "detr-tensorflow_amd")):
    sys.path.insert(0, p)
import torch

from detr_tf import _hip as hip

hip.load()
M, N, K, ak, bk = (int(x) for x in sys.argv[1:6]) if len(sys.argv) > 5 else (33600, 256, 1024, 1, 1)
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = "cuda"
torch.manual_seed(0)
A = torch.randn(M, K, device=dev) if ak else torch.randn(K, M, device=dev)
B = (torch.randn(N, K, device=dev) if bk else torch.randn(K, N, device=dev)) / K ** 0.5
C = torch.zeros(M, N, device=dev)
hip.COMPUTE_BF16 = 2
for _ in range(reps):
    hip.gemm(M, N, K, A, A.stride(0), ak, B, B.stride(0), bk, C, N, compute=2, split_k=1)
torch.cuda.synchronize()
