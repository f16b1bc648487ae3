# f32x3 GEMM timing experiments (results WRONG): libdetr_hip_x3a<bits>.so built with -DDETR_X3_ABLATE=<bits>
# 1 = one MFMA term of six, 2 = no split arithmetic, 4 = no loop requests
OUT=gpurun_out/r6_x3abl; mkdir -p $OUT
for v in base x3a1 x3a2 x3a4 x3a6; do
  if [ $v = base ]; then L=""; else L="DETR_HIP_LIB=/root/repo/detr-tensorflow_amd/lib/libdetr_hip_$v.so"; fi
  env $L python scripts/micro_split3.py $OUT/micro_$v.json > $OUT/micro_$v.log 2>&1
  echo "== $v"; grep -v amdgpu $OUT/micro_$v.log | sed -n 2,18p | cut -c1-62
done
