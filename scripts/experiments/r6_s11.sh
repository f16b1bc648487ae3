# f32x3 GEMM: 128 x 64 tiles (DETR_HIP_GEMM_TILE=2, three workgroups per CU) against the rule's tiles, per launch
OUT=gpurun_out/r6_s11; mkdir -p $OUT
timeout 300 python scripts/micro_split3.py $OUT/micro_rule.json > $OUT/micro_rule.log 2>&1; echo "== rule"; grep -v amdgpu $OUT/micro_rule.log | sed -n 1,18p | cut -c1-62
timeout 300 python scripts/micro_split3.py $OUT/micro_t2.json DETR_HIP_GEMM_TILE=2 > $OUT/micro_t2.log 2>&1; echo "== DETR_HIP_GEMM_TILE=2"; grep -v amdgpu $OUT/micro_t2.log | sed -n 1,18p | cut -c1-62
