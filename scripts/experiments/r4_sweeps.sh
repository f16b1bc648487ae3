#!/bin/bash
# round 4: re-tune the split / workgroup targets now that the partial slabs are cheap (tile-ordered, 16-byte stores)
cd /root/repo
run() { env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --launch eager --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'])"; }
run DETR_X=0
run DETR_HIP_WGRAD_FUSED_WGS=256
run DETR_HIP_WGRAD_FUSED_WGS=768
run DETR_HIP_SPLIT_TARGET=512
run DETR_HIP_SPLIT_TARGET=384
run DETR_HIP_SPLIT_TARGET64=512
run DETR_HIP_SPLIT_TARGET64=2048
run DETR_X=0
