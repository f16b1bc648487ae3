# kernel-time profile of the fp32x3 training step (rocprofv3 --kernel-trace --stats)
OUT=gpurun_out/r6_px3; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$OUT/prof -o prof -- python /root/repo/bench.py --steps 3 --warmup 2 --precision ${PREC:-fp32x3} --no-fp32-leg --no-configs --no-cpu-baseline --no-kernel-events --launch eager > /root/repo/$OUT/prof.log 2>&1); echo "prof rc=$?"
python scripts/prof_summary.py $OUT/prof/prof_results.db 3 > $OUT/prof_summary.txt 2>&1; python scripts/prof_summary.py $OUT/prof/prof_results.db 3 400 > $OUT/prof_all.txt 2>&1; head -50 $OUT/prof_summary.txt | cut -c1-200
tail -1 $OUT/prof.log | cut -c1-300
rm -rf $OUT/prof
