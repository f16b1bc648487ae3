cd /root/repo; mkdir -p gpurun_out/r5s
for rep in 1 2; do for v in on off; do
  if [ $v = off ]; then E="${AB_OFF_ENV:-DETR_HIP_GEMM_RING=2}"; else E="X=1"; fi
  env $E timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --no-kernel-events --launch eager --queries 300 --batch 16 > gpurun_out/r5s/c5_${v}_$rep.log 2>&1
  tail -1 gpurun_out/r5s/c5_${v}_$rep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5', '$v', $rep, d['ms_per_step'], d['value'])"
  env $E timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --no-kernel-events --launch eager --backbone resnet101 --height 1000 --width 1333 > gpurun_out/r5s/c4_${v}_$rep.log 2>&1
  tail -1 gpurun_out/r5s/c4_${v}_$rep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4', '$v', $rep, d['ms_per_step'], d['value'])"
done; done
