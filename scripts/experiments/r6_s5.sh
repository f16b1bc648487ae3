# f32x3 3x3 convolutions, second form (conv_x3.h): parity tests, per-launch table, fp32x3 step A/B against the first form
OUT=gpurun_out/r6_s5; mkdir -p $OUT
python -m pytest tests/test_gpu_f32x3.py -m gpu -q --no-header -p no:cacheprovider -x > $OUT/t_f32x3.log 2>&1; tail -3 $OUT/t_f32x3.log
python scripts/micro_split3.py $OUT/micro.json > $OUT/micro.log 2>&1; grep -A7 "^conv3x3" $OUT/micro.log | cut -c1-150
python scripts/micro_split3.py $OUT/micro_old.json DETR_HIP_X3_CONV=2 > $OUT/micro_old.log 2>&1; grep -A7 "^conv3x3" $OUT/micro_old.log | cut -c1-150
for v in 0 2 0 2; do
DETR_HIP_X3_CONV=$v timeout 900 python bench.py --steps 10 --warmup 3 --precision fp32x3 --no-cpu-baseline --no-configs --no-fp32-leg --launch eager > $OUT/bench_$v.log 2>&1; tail -1 $OUT/bench_$v.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('X3_CONV=$v', d['ms_per_step'], d['loss'])"
done
python -m pytest tests/test_gpu_model.py -m gpu -q --no-header -p no:cacheprovider -x -k fp32x3 > $OUT/t_model.log 2>&1; tail -3 $OUT/t_model.log
