mkdir -p gpurun_out/r6_s4
python -m pytest tests/test_gpu_f32x3.py -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/r6_s4/t_f32x3.log 2>&1; tail -3 gpurun_out/r6_s4/t_f32x3.log
python -m pytest tests/test_gpu_model.py -m gpu -q --no-header -p no:cacheprovider -x -k fp32x3 > gpurun_out/r6_s4/t_model.log 2>&1; tail -3 gpurun_out/r6_s4/t_model.log
python scripts/micro_split3.py gpurun_out/r6_s4/micro.json > gpurun_out/r6_s4/micro.log 2>&1; grep -v amdgpu gpurun_out/r6_s4/micro.log | cut -c1-70
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs --launch eager > gpurun_out/r6_s4/bench.log 2>&1; tail -1 gpurun_out/r6_s4/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['fp32']['ms_per_step'], d['fp32'].get('f32x3',{}).get('ms_per_step'))"
