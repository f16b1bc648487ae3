import os, sys
os.environ["TUNE_ACT16"] = "1"
sys.argv = ["tune_gemm.py", "none"]
sys.path.insert(0, "/root/repo/scripts")
import importlib.util
spec = importlib.util.spec_from_file_location("tg", "/root/repo/scripts/tune_gemm.py")
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
tg.hip.COMPUTE_BF16 = 1
for (M, N, K, ak, bk, res) in [(534400, 256, 64, 1, 0, True), (534400, 256, 64, 1, 1, True), (133600, 512, 128, 1, 0, True), (33600, 1024, 256, 1, 0, True),
                               (534400, 64, 256, 1, 1, False), (133600, 128, 512, 1, 1, False), (33600, 256, 1024, 1, 0, False)]:
    tg.gemm_case(M, N, K, ak, bk, res=res, tiles=(2, 3))
