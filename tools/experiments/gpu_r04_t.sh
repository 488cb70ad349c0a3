#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_t; mkdir -p $OUT
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
for n in reforder fastscale; do
  [ $n = fastscale ] && cp tools/experiments/bin/libedgehip_fastscale.so rebvo_amd/lib/libedgehip.so
  echo "== $n 1280x720"; timeout 300 python tools/experiments/exp_pipeline_closeness.py 1280 720 5 2>&1 | tail -14
  echo "== $n 752x480"; timeout 300 python tools/experiments/exp_pipeline_closeness.py 752 480 8 2>&1 | tail -14
done | tee $OUT/closeness.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
