"""Probe: would deterministic split-K (+ a reduce launch) pay for the long-K, small-M forward / dgrad GEMMs of the
transformer FFN (M = 800 / 8400, N = 256, K = 2048)?  Times main + reduce launches together, no epilogue."""
import os, sys, importlib.util
sys.argv = ["tune_gemm.py", "none"]
spec = importlib.util.spec_from_file_location("tg", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tune_gemm.py"))
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
tg.hip.COMPUTE_BF16 = 1
for (M, N, K, ak, bk) in [(800, 256, 2048, 1, 1), (800, 256, 2048, 1, 0), (8400, 256, 2048, 1, 1), (8400, 256, 2048, 1, 0), (800, 2048, 256, 1, 1), (800, 256, 256, 1, 1)]:
    tg.gemm_case(M, N, K, ak, bk, splits=(1, 2, 4, 8, 16), tiles=(3,))
