#!/bin/bash
# the driver's launch form (torch.distributed.run) of the 2-rank gloo bench on one GPU; after PROBE_T seconds SIGABRT goes to the whole process group and
# faulthandler (PYTHONFAULTHANDLER=1) prints every rank's Python stack: where does a hang sit?
cd /root/repo
OUT=gpurun_out/r5_dp2probe; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 PYTHONUNBUFFERED=1
setsid python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --batch 2 --height 256 --width 320 --dist-backend gloo --no-cpu-baseline --no-kernel-events > $OUT/torchrun.log 2>&1 &
PID=$!
for i in $(seq 1 ${PROBE_T:-120}); do sleep 1; kill -0 $PID 2>/dev/null || break; done
if kill -0 $PID 2>/dev/null; then echo "still running after ${PROBE_T:-120}s: SIGABRT to the group"; kill -ABRT -- -$PID; sleep 3; kill -KILL -- -$PID 2>/dev/null; else echo "finished by itself"; fi
grep -v "amdgpu.ids\|socket.cpp" $OUT/torchrun.log | tail -${PROBE_TAIL:-60} | cut -c1-200
