#!/bin/bash
# Builds timing-experiment variants of csrc/attention_dma.hip (A2_ABLATE bits, see the source) into detr-tensorflow_amd/lib/alt/
# (git-ignored; they travel to the GPU box with the snapshot).  Results of those libraries are WRONG by construction.
# usage: scripts/experiments/attn2_ablate.sh 1 2 4 ...      then on the GPU box:  DETR_HIP_LIB=.../lib/alt/libdetr_hip_a2ab<N>.so python scripts/micro_attn2.py
set -e
cd "$(dirname "$0")/../../detr-tensorflow_amd"
mkdir -p build_alt lib/alt
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -DA2_ABLATE=$n -c csrc/attention_dma.hip -o build_alt/attention_dma_ab$n.o 2>/dev/null
  objs=$(ls build/*.o | grep -v attention_dma.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_alt/attention_dma_ab$n.o -o lib/alt/libdetr_hip_a2ab$n.so
  echo "built lib/alt/libdetr_hip_a2ab$n.so"
done
