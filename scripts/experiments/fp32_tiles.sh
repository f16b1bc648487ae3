cd /root/repo
for t in 0 2 1; do
  DETR_HIP_GEMM_TILE=$t timeout 600 python bench.py --steps 10 --warmup 3 --precision fp32 --no-cpu-baseline --no-fp32-leg --no-configs --launch eager --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile', $t, d['ms_per_step'], d['value'])"
done
