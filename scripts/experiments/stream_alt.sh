#!/bin/bash
# streaming kernel micro: the library in place vs lib/libdetr_hip_alt.so
cd /root/repo
for v in new alt; do
  [ $v = alt ] && export DETR_HIP_LIB=/root/repo/detr-tensorflow_amd/lib/libdetr_hip_alt.so
  echo "== $v"; timeout 200 python scripts/experiments/ablate_stream.py 2>&1 | tail -1 | tr "|" "\n"
done
