#!/bin/bash
# Ablation timing of the bf16 tile kernels (common.h DETR_ABLATE bits): which resource sets their time?
#   scripts/experiments/ablate.sh build   (CPU: builds lib/ablate/libdetr_hip_a<bits>.so, only gemm_f32 / conv_f32 are recompiled)
#   scripts/experiments/ablate.sh run     (GPU: times every build with ablate_time.py)
cd /root/repo/detr-tensorflow_amd
VARIANTS="${ABLATE_VARIANTS:-0 1 2 4 8 16 6 22 30 31}"
FILES="${ABLATE_FILES:-gemm_f32 conv_f32}"       # sources recompiled with -DDETR_ABLATE (the others are reused)
if [ "$1" = build ]; then
  mkdir -p lib/ablate build/ablate
  for v in $VARIANTS; do
    ( for f in $FILES; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -DDETR_ABLATE=$v -c csrc/$f.hip -o build/ablate/${f}_a$v.o &
      done
      wait
      OBJ=$(ls build/*.o | grep -v "$(echo $FILES | sed 's/ /.o\\|/g').o")
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ $(for f in $FILES; do echo build/ablate/${f}_a$v.o; done) -o lib/ablate/libdetr_hip_a$v.so
      echo built $v ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait
else
  for v in $VARIANTS; do
    echo "== DETR_ABLATE=$v"; DETR_HIP_LIB=/root/repo/detr-tensorflow_amd/lib/ablate/libdetr_hip_a$v.so timeout 300 python /root/repo/scripts/experiments/ablate_time.py 2>&1 | tail -2
  done
fi
