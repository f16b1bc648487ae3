#!/bin/bash
# same-box A/B of one environment switch: scripts/experiments/envab.sh NAME=VALUE [tag]
cd /root/repo
run() { env "$1" timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --launch eager --dump-shapes gpurun_out/envab_$2.json 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['value'])"; }
run DETR_X=0 new
run "$1" env
run DETR_X=0 new2
run "$1" env2
