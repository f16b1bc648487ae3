#!/bin/bash
# same-box A/B of one environment switch: usage ab_env.sh VAR ON OFF   (bench without kernel events, alternating)
cd /root/repo
VAR=$1; ON=$2; OFF=$3
for rep in 1 2; do
  for v in $ON $OFF; do
    env $VAR=$v python bench.py --steps 6 --warmup 2 --precision bf16 --no-fp32-leg --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['value'])"
  done
done
