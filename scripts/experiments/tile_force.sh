#!/bin/bash
# per-shape effect of forcing one tile shape on every bf16 tile-engine GEMM (DETR_HIP_GEMM_TILE: 2 = 128x64, 1 = 128x128, 3 = 64x64, 5 = 64x128)
cd /root/repo
for t in 0 2 1 5; do
  DETR_HIP_GEMM_TILE=$t timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --launch eager --dump-shapes gpurun_out/tileforce_$t.json 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile', $t, d['ms_per_step'], d['value'])"
done
