#!/bin/bash
# same-box A/B of two builds of libdetr_hip.so: NEW (in place) vs PREV (lib/libdetr_hip_prev.so), alternating
cd /root/repo
L=detr-tensorflow_amd/lib
cp $L/libdetr_hip.so /tmp/new.so; cp $L/libdetr_hip_prev.so /tmp/prev.so
for rep in 1 2; do
  for v in new prev; do
    cp /tmp/$v.so $L/libdetr_hip.so
    python bench.py --steps 6 --warmup 2 --precision bf16 --no-fp32-leg --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
  done
done
cp /tmp/new.so $L/libdetr_hip.so
