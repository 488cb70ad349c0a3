#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_u; mkdir -p $OUT
for sz in "1024 576" "1280 720" "1000 480" "752 720" "1280 480"; do
  echo "== $sz"; timeout 300 python tools/experiments/exp_pipeline_closeness.py $sz 4 2>&1 | tail -5
done | tee $OUT/closeness.txt
