#!/bin/bash
# Runs on the GPU box via gpurun: tests, smoke, bench; logs under gpurun_out/.
# usage: scripts/gpu_suite.sh [tag] [what...]   what in: kernels model smoke bench fwd prof
TAG=${1:-run}; shift
WHAT=${@:-kernels model smoke bench}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /root/repo
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
for w in $WHAT; do
  case $w in
    kernels) timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --no-header -p no:cacheprovider > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/summary.txt ;;
    kernels_all) timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/summary.txt ;;
    model) timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --no-header -p no:cacheprovider > $OUT/model.log 2>&1; echo "model rc=$?" >> $OUT/summary.txt ;;
    debug) timeout 300 python scripts/debug_grad.py > $OUT/debug.log 2>&1; echo "debug rc=$?" >> $OUT/summary.txt; tail -30 $OUT/debug.log ;;
    golden) timeout 300 python -m pytest tests/test_golden.py -m gpu -q --no-header -p no:cacheprovider > $OUT/golden.log 2>&1; echo "golden rc=$?" >> $OUT/summary.txt; tail -15 $OUT/golden.log ;;
    dp) timeout 600 python -m pytest tests/test_gpu_dp.py -m gpu -q --no-header -p no:cacheprovider > $OUT/dp.log 2>&1; echo "dp rc=$?" >> $OUT/summary.txt; tail -12 $OUT/dp.log
        timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --dist-backend gloo > $OUT/bench_dp2_gloo.log 2>&1; echo "bench dp2 rc=$?" >> $OUT/summary.txt; tail -3 $OUT/bench_dp2_gloo.log ;;
    nccl1) DETR_DP_FORCE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 3 --warmup 1 --precision bf16 --no-cpu-baseline > $OUT/bench_nccl1.log 2>&1; echo "bench nccl ws1 rc=$?" >> $OUT/summary.txt; tail -3 $OUT/bench_nccl1.log | cut -c1-600 ;;
    smoke)timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt ;;
    bench) timeout 600 python bench.py --steps 5 --warmup 2 --dump-shapes $OUT/shapes.json > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/summary.txt ;;
    bench32) timeout 600 python bench.py --steps 5 --warmup 2 --precision fp32 --no-cpu-baseline --dump-shapes $OUT/shapes_f32.json > $OUT/bench_f32.log 2>&1; echo "bench32 rc=$?" >> $OUT/summary.txt; tail -1 $OUT/bench_f32.log | cut -c1-400 ;;
    bench16) timeout 600 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline --dump-shapes $OUT/shapes_bf16.json > $OUT/bench_bf16.log 2>&1; echo "bench16 rc=$?" >> $OUT/summary.txt; tail -1 $OUT/bench_bf16.log | cut -c1-400 ;;
    fwd) timeout 300 python bench.py --steps 5 --warmup 2 --mode fwdloss --no-cpu-baseline > $OUT/bench_fwd.log 2>&1; echo "fwd rc=$?" >> $OUT/summary.txt ;;
    prof) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof -o prof -- python /root/repo/bench.py --steps 3 --warmup 1 --precision fp32 --no-cpu-baseline --no-kernel-events > /root/repo/$OUT/prof.log 2>&1); echo "prof rc=$?" >> $OUT/summary.txt
          python scripts/prof_summary.py $OUT/prof/prof_results.db 3 > $OUT/prof_summary.txt 2>&1 ;;
    prof16) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof16 -o prof -- python /root/repo/bench.py --steps 3 --warmup 1 --precision bf16 --no-fp32-leg --no-cpu-baseline --no-kernel-events > /root/repo/$OUT/prof16.log 2>&1); echo "prof16 rc=$?" >> $OUT/summary.txt
          python scripts/prof_summary.py $OUT/prof16/prof_results.db 3 > $OUT/prof16_summary.txt 2>&1 ;;
    sq) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d /root/repo/$OUT/pmc_SQ -o pmc -- python /root/repo/bench.py --steps 2 --warmup 1 --precision ${PMC_PREC:-fp32} --no-fp32-leg --no-cpu-baseline --no-kernel-events > /root/repo/$OUT/pmc_SQ.log 2>&1); echo "pmc SQ rc=$?" >> $OUT/summary.txt
          python scripts/pmc_sq_summary.py $OUT > $OUT/pmc_sq_summary.txt 2>&1; head -30 $OUT/pmc_sq_summary.txt ;;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c -d /root/repo/$OUT/pmc_$c -o pmc -- python /root/repo/bench.py --steps 2 --warmup 1 --precision ${PMC_PREC:-fp32} --no-fp32-leg --no-cpu-baseline --no-kernel-events > /root/repo/$OUT/pmc_$c.log 2>&1); echo "pmc $c rc=$?" >> $OUT/summary.txt; done
          python scripts/pmc_summary.py $OUT ${PMC_PAT:-gemm_f32_kernel} $OUT/traffic.json > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt | head -30 ;;
  esac
done
cat $OUT/summary.txt
for f in kernels model smoke bench bench_fwd; do [ -f $OUT/$f.log ] && { echo "=== $f"; tail -n 40 $OUT/$f.log; }; done
exit 0
