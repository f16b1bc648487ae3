#!/bin/bash
# Round-2 GPU runner (via gpurun): scripts/gpu_round2.sh <tag> <what...>
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /root/repo
for w in "$@"; do
  case $w in
    tests) timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.txt; tail -40 $OUT/tests.log ;;
    tests_all) timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.txt; tail -60 $OUT/tests.log ;;
    kernels) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/summary.txt; tail -40 $OUT/kernels.log ;;
    model) timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_training.py tests/test_golden.py tests/test_gpu_dp.py -m gpu -q --no-header -p no:cacheprovider > $OUT/model.log 2>&1; echo "model rc=$?" >> $OUT/summary.txt; tail -60 $OUT/model.log ;;
    smoke) timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt; tail -3 $OUT/smoke.log ;;
    bench) timeout 900 python bench.py --steps 10 --warmup 3 --dump-shapes $OUT/shapes.json > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/summary.txt; tail -2 $OUT/bench.log | cut -c1-1500 ;;
    bench_quick) timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --dump-shapes $OUT/shapes.json > $OUT/bench_quick.log 2>&1; echo "bench_quick rc=$?" >> $OUT/summary.txt; tail -2 $OUT/bench_quick.log | cut -c1-2500 ;;
    bench_eager) timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-graph --no-kernel-events > $OUT/bench_eager.log 2>&1; echo "bench_eager rc=$?" >> $OUT/summary.txt; tail -2 $OUT/bench_eager.log | cut -c1-600 ;;
    bench_f32) timeout 600 python bench.py --steps 5 --warmup 2 --precision fp32 --no-cpu-baseline --dump-shapes $OUT/shapes_f32.json > $OUT/bench_f32.log 2>&1; echo "bench_f32 rc=$?" >> $OUT/summary.txt; tail -2 $OUT/bench_f32.log | cut -c1-800 ;;
    prof16) (cd /tmp && export TMPDIR=/tmp DETR_HIP_WGRAD_STREAM=0 && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof16 -o prof -- python /root/repo/bench.py --steps 3 --warmup 2 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --no-graph > /root/repo/$OUT/prof16.log 2>&1); echo "prof16 rc=$?" >> $OUT/summary.txt
          python scripts/prof_summary.py $OUT/prof16/prof_results.db 3 > $OUT/prof16_summary.txt 2>&1; head -70 $OUT/prof16_summary.txt
          python scripts/launch_count.py $OUT/prof16/prof_results.db 10 > $OUT/launch_count.txt 2>&1; cat $OUT/launch_count.txt; rm -rf $OUT/prof16 ;;
    n4) timeout 600 python -m pytest tests/test_gpu_model.py -k "tf_backbone" tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider > $OUT/n4.log 2>&1; echo "n4 rc=$?" >> $OUT/summary.txt; tail -30 $OUT/n4.log ;;
    ab_split) for v in 1 2; do DETR_HIP_SPLIT_XCD=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --dump-shapes $OUT/shapes_split$v.json > $OUT/bench_split$v.log 2>&1; echo "split$v rc=$?" >> $OUT/summary.txt; tail -1 $OUT/bench_split$v.log | cut -c1-330; done ;;
    ab_defer) for v in "0 512" "1 512" "1 128"; do set -- $v; DETR_HIP_DEFER_REDUCE=$1 DETR_HIP_DEFER_MB=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --dump-shapes $OUT/shapes_defer$1_$2.json > $OUT/bench_defer$1_$2.log 2>&1; echo "defer$1_$2 rc=$?" >> $OUT/summary.txt; tail -1 $OUT/bench_defer$1_$2.log | cut -c1-330; done ;;
    listpmc) (cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 -L > /root/repo/$OUT/pmc_avail.txt 2>&1); grep -c "SQ_" $OUT/pmc_avail.txt ;;
    pmc16) for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export TMPDIR=/tmp DETR_HIP_WGRAD_STREAM=0 && timeout 600 rocprofv3 --pmc $c -d /root/repo/$OUT/pmc_$c -o pmc -- python /root/repo/bench.py --steps 2 --warmup 2 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --no-graph > /root/repo/$OUT/pmc_$c.log 2>&1); echo "pmc $c rc=$?" >> $OUT/summary.txt; done
          python scripts/pmc_summary.py $OUT gemm_stream $OUT/traffic_bf16.json > $OUT/pmc_hbm_summary.txt 2>&1; head -30 $OUT/pmc_hbm_summary.txt; rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE ;;
    pmc32) for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc $c -d /root/repo/$OUT/pmc_$c -o pmc -- python /root/repo/bench.py --steps 1 --warmup 1 --precision fp32 --no-configs --no-cpu-baseline --no-kernel-events --no-graph > /root/repo/$OUT/pmc32_$c.log 2>&1); echo "pmc32 $c rc=$?" >> $OUT/summary.txt; done
          python scripts/pmc_summary.py $OUT gemm_f32 $OUT/traffic.json > $OUT/pmc_hbm_summary_fp32.txt 2>&1; head -12 $OUT/pmc_hbm_summary_fp32.txt; rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE ;;
    sqA) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d /root/repo/$OUT/pmc_SQ -o pmc -- python /root/repo/bench.py --steps 2 --warmup 2 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --no-graph > /root/repo/$OUT/pmc_SQ.log 2>&1); echo "pmc SQ rc=$?" >> $OUT/summary.txt
          python scripts/pmc_sq_summary.py $OUT pmc_SQ > $OUT/pmc_sq_summary.txt 2>&1; head -40 $OUT/pmc_sq_summary.txt | cut -c1-120; rm -rf $OUT/pmc_SQ ;;
    sqB) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d /root/repo/$OUT/pmc_SQB -o pmc -- python /root/repo/bench.py --steps 2 --warmup 2 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --no-graph > /root/repo/$OUT/pmc_SQB.log 2>&1); echo "pmc SQB rc=$?" >> $OUT/summary.txt
          python scripts/pmc_sq_summary.py $OUT pmc_SQB > $OUT/pmc_sqb_summary.txt 2>&1; head -5 $OUT/pmc_sqb_summary.txt | cut -c1-120; rm -rf $OUT/pmc_SQB ;;
    *) echo "unknown $w" ;;
  esac
done
cat $OUT/summary.txt
exit 0
