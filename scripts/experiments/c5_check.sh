cd /root/repo
run() { env "$@" timeout 900 python bench.py --queries 300 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --no-kernel-events 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['value'], d['config']['launch'][:90])"; }
run DETR_X=0
run DETR_HIP_MASK_BITS=0
run DETR_HIP_SLAB_TS=2
run DETR_HIP_LIB=/root/repo/detr-tensorflow_amd/lib/libdetr_hip_alt.so
