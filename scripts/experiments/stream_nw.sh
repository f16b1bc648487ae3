#!/bin/bash
# K = 256 streaming kernel: one (DETR_HIP_STREAM_NW=1) vs two (=2) column slices per wave, then the rule (=0)
cd /root/repo
for m in 1 2 3; do echo "== DETR_HIP_STREAM_NW=$m"; DETR_HIP_STREAM_NW=$m timeout 200 python scripts/experiments/ablate_stream.py 2>&1 | tail -1 | tr "|" "\n"; done
