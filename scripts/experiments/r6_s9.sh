# f32x3 3x3 convolution: 128 x 64 tiles (3 workgroups per CU) against the rule's tiles, per launch
OUT=gpurun_out/r6_s9; mkdir -p $OUT
for v in 0 3; do
timeout 300 python scripts/micro_split3.py $OUT/micro_$v.json DETR_HIP_X3_T192=$v > $OUT/micro_$v.log 2>&1; echo "== DETR_HIP_X3_T192=$v"; grep -A7 "^conv3x3" $OUT/micro_$v.log | cut -c1-110
done
