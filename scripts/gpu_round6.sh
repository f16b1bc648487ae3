#!/bin/bash
# Round-6 GPU runner (via gpurun): scripts/gpu_round6.sh <tag> <what...>
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PYTHONUNBUFFERED=1
cd /root/repo
for w in "$@"; do
  case $w in
    tests) timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.txt; tail -40 $OUT/tests.log ;;
    tests_all) timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.txt; tail -80 $OUT/tests.log ;;
    tests_new) timeout 1500 python -m pytest tests/test_gpu_configs_full.py tests/test_gpu_training.py -m gpu -q -s --no-header -p no:cacheprovider > $OUT/tests_new.log 2>&1; echo "tests_new rc=$?" >> $OUT/summary.txt; grep -E "^\[|passed|failed|Error|assert" $OUT/tests_new.log | cut -c1-600 | tail -40 ;;
    kernels) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider > $OUT/kernels.log 2>&1; echo "kernels rc=$?" >> $OUT/summary.txt; tail -40 $OUT/kernels.log ;;
    kfast) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider -x -k "${KEXPR:-gemm}" > $OUT/kfast.log 2>&1; echo "kfast rc=$?" >> $OUT/summary.txt; tail -30 $OUT/kfast.log ;;
    kring) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider -k "ring" > $OUT/kring.log 2>&1; echo "kring rc=$?" >> $OUT/summary.txt; grep -E "passed|failed|FAILED|Error|assert " $OUT/kring.log | cut -c1-300 | tail -40 ;;
    mconv) timeout 600 python scripts/micro_conv.py ${MCONV_ARGS:-DETR_HIP_CONV_DMA=2} > $OUT/mconv.log 2>&1; echo "mconv rc=$?" >> $OUT/summary.txt; grep -v amdgpu.ids $OUT/mconv.log | cut -c1-200 | tail -30 ;;
    mring) timeout 900 python scripts/micro_ring.py $OUT/mring.json ${MRING_ARGS:--} > $OUT/mring.log 2>&1; echo "mring rc=$?" >> $OUT/summary.txt; grep -v amdgpu.ids $OUT/mring.log | tail -30 ;;
    rsweep) timeout 1200 python scripts/experiments/ring_sweep.py $OUT/ring_sweep.json ${RSWEEP_ARGS:-} > $OUT/ring_sweep.log 2>&1; echo "rsweep rc=$?" >> $OUT/summary.txt; grep -v amdgpu.ids $OUT/ring_sweep.log | cut -c1-330 | tail -30 ;;
    model) timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_training.py tests/test_golden.py tests/test_gpu_dp.py -m gpu -q --no-header -p no:cacheprovider > $OUT/model.log 2>&1; echo "model rc=$?" >> $OUT/summary.txt; tail -60 $OUT/model.log ;;
    smoke) timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.txt; tail -3 $OUT/smoke.log ;;
    bench) timeout 900 python bench.py --steps 20 --warmup 3 --dump-shapes $OUT/shapes.json > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/summary.txt; tail -1 $OUT/bench.log > $OUT/bench_line.json; tail -2 $OUT/bench.log | cut -c1-1800 ;;
    ab_split) for v in ${AB_SPLITS:-256,1024 128,1024 256,512 384,1024}; do DETR_HIP_SPLIT_TARGET=${v%,*} DETR_HIP_SPLIT_TARGET64=${v#*,} timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --dump-shapes "$OUT/shapes_split_$v.json" > "$OUT/bench_split_$v.log" 2>&1; echo "split $v rc=$?" >> $OUT/summary.txt; tail -n 1 "$OUT/bench_split_$v.log" | cut -c1-330; done ;;
    bench_quick) timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --dump-shapes $OUT/shapes.json > $OUT/bench_quick.log 2>&1; echo "bench_quick rc=$?" >> $OUT/summary.txt; tail -1 $OUT/bench_quick.log > $OUT/bench_quick_line.json; tail -2 $OUT/bench_quick.log | cut -c1-1200 ;;
    bench_dp2) timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --batch 2 --height 256 --width 320 --dist-backend gloo --no-cpu-baseline --no-kernel-events > $OUT/bench_dp2_gloo.log 2>&1; echo "bench dp2 rc=$?" >> $OUT/summary.txt; tail -1 $OUT/bench_dp2_gloo.log | cut -c1-1500 ;;
    micro) timeout 900 python scripts/micro_gemm.py $OUT/micro.json ${MICRO_ARGS:-} > $OUT/micro.log 2>&1; echo "micro rc=$?" >> $OUT/summary.txt; tail -70 $OUT/micro.log ;;
    micro_cold) timeout 900 python scripts/micro_gemm.py $OUT/micro_cold.json --cold ${MICRO_ARGS:-} > $OUT/micro_cold.log 2>&1; echo "micro_cold rc=$?" >> $OUT/summary.txt; tail -70 $OUT/micro_cold.log ;;
    micro_wgrad) timeout 900 python scripts/micro_wgrad.py $OUT/micro_wgrad.json > $OUT/micro_wgrad.log 2>&1; echo "micro_wgrad rc=$?" >> $OUT/summary.txt; grep -v amdgpu $OUT/micro_wgrad.log | cut -c1-400 ;;
    micro_attn) timeout 600 python scripts/micro_attn.py > $OUT/micro_attn.log 2>&1; echo "micro_attn rc=$?" >> $OUT/summary.txt; tail -12 $OUT/micro_attn.log ;;
    prof16) (cd /tmp && export TMPDIR=/tmp DETR_HIP_WGRAD_STREAM=0 && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof16 -o prof -- python /root/repo/bench.py --steps 3 --warmup 2 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --launch eager > /root/repo/$OUT/prof16.log 2>&1); echo "prof16 rc=$?" >> $OUT/summary.txt
          python scripts/prof_summary.py $OUT/prof16/prof_results.db 3 > $OUT/prof16_summary.txt 2>&1; python scripts/prof_summary.py $OUT/prof16/prof_results.db 3 400 > $OUT/prof16_all.txt 2>&1; head -70 $OUT/prof16_summary.txt
          python scripts/launch_count.py $OUT/prof16/prof_results.db 10 > $OUT/launch_count.txt 2>&1; cat $OUT/launch_count.txt; rm -rf $OUT/prof16 ;;
    timeline) for l in eager graph; do (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d /root/repo/$OUT/tl_$l -o prof -- python /root/repo/bench.py --steps 4 --warmup 8 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --launch $l > /root/repo/$OUT/tl_$l.log 2>&1); echo "timeline $l rc=$?" >> $OUT/summary.txt; done
          python scripts/prof_timeline2.py $(ls $OUT/tl_eager/*/*.db $OUT/tl_eager/*.db 2>/dev/null | head -1) $(ls $OUT/tl_graph/*/*.db $OUT/tl_graph/*.db 2>/dev/null | head -1) > $OUT/timeline_graph_vs_eager.txt 2>&1; cat $OUT/timeline_graph_vs_eager.txt | cut -c1-400; rm -rf $OUT/tl_eager $OUT/tl_graph ;;
    pmc16) for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export TMPDIR=/tmp DETR_HIP_WGRAD_STREAM=0 && timeout 600 rocprofv3 --pmc $c -d /root/repo/$OUT/pmc_$c -o pmc -- python /root/repo/bench.py --steps 2 --warmup 2 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --launch eager > /root/repo/$OUT/pmc_$c.log 2>&1); echo "pmc $c rc=$?" >> $OUT/summary.txt; done
          python scripts/pmc_summary.py $OUT gemm_bf16c $OUT/traffic_bf16.json > $OUT/pmc_hbm_summary.txt 2>&1; head -30 $OUT/pmc_hbm_summary.txt; rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE ;;
    sqA) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d /root/repo/$OUT/pmc_SQ -o pmc -- python /root/repo/bench.py --steps 2 --warmup 2 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --launch eager > /root/repo/$OUT/pmc_SQ.log 2>&1); echo "pmc SQ rc=$?" >> $OUT/summary.txt
          python scripts/pmc_sq_summary.py $OUT pmc_SQ > $OUT/pmc_sq_summary.txt 2>&1; head -40 $OUT/pmc_sq_summary.txt | cut -c1-120; rm -rf $OUT/pmc_SQ ;;
    ab2) # same-box A/B: new library | lib/libdetr_hip_alt.so (whatever the alternative build of the moment is)
         for rep in 1 2; do
           for v in new alt; do
             case $v in
               new) envs="" ;;
               alt) envs="DETR_HIP_LIB=/root/repo/detr-tensorflow_amd/lib/libdetr_hip_alt.so" ;;
             esac
             env $envs timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --launch eager --dump-shapes $OUT/shapes_$v.json > $OUT/ab_${v}_$rep.log 2>&1
             echo "ab $v $rep rc=$?" >> $OUT/summary.txt
             tail -1 $OUT/ab_${v}_$rep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', $rep, d['ms_per_step'], d['value'], d['loss'])"
           done
         done ;;
    abenv) # same-box A/B of an environment switch: AB_ENV="NAME=VALUE" (the B arm), default arm first
         for rep in 1 2; do
           for v in new env; do
             case $v in
               new) envs="" ;;
               env) envs="$AB_ENV" ;;
             esac
             env $envs timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --launch eager --dump-shapes $OUT/shapes_$v.json > $OUT/ab_${v}_$rep.log 2>&1
             echo "ab $v $rep rc=$?" >> $OUT/summary.txt
             tail -1 $OUT/ab_${v}_$rep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', $rep, d['ms_per_step'], d['value'], d['loss'])"
           done
         done ;;
    ab3) # same-box A/B/C: new library | compiler-scheduled K loop (lib/libdetr_hip_nopipe.so) | row-major split-K slabs
         for rep in 1 2; do
           for v in new nopipe rowmajor; do
             case $v in
               new) envs="" ;;
               nopipe) envs="DETR_HIP_LIB=/root/repo/detr-tensorflow_amd/lib/libdetr_hip_nopipe.so" ;;
               rowmajor) envs="DETR_HIP_SLAB_TS=2" ;;
             esac
             env $envs timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --launch eager --dump-shapes $OUT/shapes_$v.json > $OUT/ab_${v}_$rep.log 2>&1
             echo "ab $v $rep rc=$?" >> $OUT/summary.txt
             tail -1 $OUT/ab_${v}_$rep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', $rep, d['ms_per_step'], d['value'], d['loss'])"
           done
         done ;;
    timeline) (cd /tmp && export TMPDIR=/tmp DETR_HIP_WGRAD_STREAM=${TL_STREAMS:-1} && timeout 600 rocprofv3 --kernel-trace -d /root/repo/$OUT/proftl -o prof -- python /root/repo/bench.py --steps 3 --warmup 2 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --launch eager > /root/repo/$OUT/proftl.log 2>&1); echo "timeline rc=$?" >> $OUT/summary.txt
          python scripts/prof_timeline.py $OUT/proftl/prof_results.db 1 > $OUT/timeline.txt 2>&1; head -3 $OUT/timeline.txt; rm -rf $OUT/proftl ;;
    timeline_graph) (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace -d /root/repo/$OUT/proftlg -o prof -- python /root/repo/bench.py --steps 3 --warmup 3 --precision bf16 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --launch graph > /root/repo/$OUT/proftlg.log 2>&1); echo "timeline_graph rc=$?" >> $OUT/summary.txt
          python scripts/prof_timeline.py $OUT/proftlg/prof_results.db 1 > $OUT/timeline_graph.txt 2>&1; head -3 $OUT/timeline_graph.txt; rm -rf $OUT/proftlg ;;
    phases) timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-configs --no-kernel-events --launch ${PH_LAUNCH:-eager} --phase-events > $OUT/phases.log 2>&1; echo "phases rc=$?" >> $OUT/summary.txt
          tail -1 $OUT/phases.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step']); print(json.dumps(d['phases_ms'], indent=1))" ;;
    pmc32) for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export TMPDIR=/tmp DETR_HIP_WGRAD_STREAM=0 && timeout 900 rocprofv3 --pmc $c -d /root/repo/$OUT/pmc_$c -o pmc -- python /root/repo/bench.py --steps 2 --warmup 2 --precision fp32 --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --launch eager > /root/repo/$OUT/pmc32_$c.log 2>&1); echo "pmc32 $c rc=$?" >> $OUT/summary.txt; done
          python scripts/pmc_summary.py $OUT gemm_f32 $OUT/traffic_fp32.json > $OUT/pmc_hbm_fp32_summary.txt 2>&1; head -30 $OUT/pmc_hbm_fp32_summary.txt | cut -c1-170; rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE ;;
    benchfull) timeout 1200 python bench.py --dump-shapes $OUT/shapes.json --phase-events > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/summary.txt; tail -1 $OUT/bench.log > $OUT/bench_line.json; tail -1 $OUT/bench.log | cut -c1-2500 ;;
    tests_r4) timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --maxfail=12 -s > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/summary.txt; grep -E "^\[|passed|failed|FAILED|Error" $OUT/tests.log | cut -c1-400 | tail -60 ;;
    *) echo "unknown $w" ;;
  esac
done
cat $OUT/summary.txt
exit 0
