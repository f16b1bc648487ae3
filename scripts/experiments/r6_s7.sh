# f32x3 GEMM, 192 x 128 tiles: parity tests, per-launch table with / without, fp32x3 step A/B
OUT=gpurun_out/r6_s10; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_f32x3.py -m gpu -q --no-header -p no:cacheprovider -x > $OUT/t_f32x3.log 2>&1; tail -3 $OUT/t_f32x3.log
timeout 300 python scripts/micro_split3.py $OUT/micro.json > $OUT/micro.log 2>&1; grep -A7 "^conv3x3" $OUT/micro.log | cut -c1-150
timeout 300 python scripts/micro_split3.py $OUT/micro_off.json DETR_HIP_X3_T192=2 > $OUT/micro_off.log 2>&1; grep -A7 "^conv3x3" $OUT/micro_off.log | cut -c1-150
for v in 0 2 0 2; do
DETR_HIP_X3_T192=$v timeout 900 python bench.py --steps 10 --warmup 3 --precision fp32x3 --no-cpu-baseline --no-configs --no-fp32-leg --launch eager > $OUT/bench_$v.log 2>&1; tail -1 $OUT/bench_$v.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('X3_T192=$v', d['ms_per_step'], d['loss'])"
done
