# f32x3 weight gradients: round-aware split plans (GEMM: DETR_HIP_X3_SPLIT_TARGET, 3x3: DETR_HIP_X3_WG_ROUNDS) per launch
OUT=gpurun_out/r6_s12; mkdir -p $OUT
for v in "512 3" "0 3" "1024 3" "512 2" "512 1"; do set -- $v
DETR_HIP_X3_SPLIT_TARGET=$1 timeout 300 python scripts/micro_split3.py $OUT/micro_$1_$2.json DETR_HIP_X3_WG_ROUNDS=$2 > $OUT/micro_$1_$2.log 2>&1
echo "== X3_SPLIT_TARGET=$1 X3_WG_ROUNDS=$2"; grep -v amdgpu $OUT/micro_$1_$2.log | grep "^wgrad" | cut -c1-62; grep -A6 "^conv3x3" $OUT/micro_$1_$2.log | cut -c1-34,75-100
done
