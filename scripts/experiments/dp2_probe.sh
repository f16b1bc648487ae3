#!/bin/bash
# two ranks of bench.py on ONE GPU over gloo, each under `timeout -s ABRT` with faulthandler: where does a hang sit?
cd /root/repo
OUT=gpurun_out/r5_dp2probe; mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 PYTHONUNBUFFERED=1
ARGS="--gpus 2 --steps 4 --warmup 1 --batch 2 --height 256 --width 320 --dist-backend gloo --no-cpu-baseline --no-kernel-events ${EXTRA_ARGS:-}"
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT ${PROBE_T:-150} python -X faulthandler bench.py $ARGS > $OUT/rank$r.log 2>&1 &
done
wait
for r in 0 1; do echo "== rank $r"; grep -v "amdgpu.ids\|socket.cpp" $OUT/rank$r.log | tail -${PROBE_TAIL:-45} | cut -c1-220; done
