#!/bin/bash
# PMC pass over the stream-vs-generic timing script: FETCH_SIZE / WRITE_SIZE per kernel (KB units; gfx950 FETCH x2)
OUT=gpurun_out/r4l; mkdir -p $OUT
cd /root/repo
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc $c -d /root/repo/$OUT/pmc_$c -o pmc -- python /root/repo/scripts/experiments/stream_vs_generic.py > /root/repo/$OUT/pmc_$c.log 2>&1)
done
python scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
rm -f $OUT/pmc_*/*.db $OUT/pmc_*/*/*.db
head -40 $OUT/pmc_summary.txt
