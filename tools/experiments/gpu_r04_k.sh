#!/bin/bash
# round 4, call k: the one-kernel stage A with its fast ticks (row tests folded) and branch-free edge / publish stores: parity, then A/B
set -u
OUT=$PWD/gpurun_out/r04_k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_fused_stage_a_gpu.py tests/test_grey8_gpu.py tests/test_undistort_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -4
ab() {
  echo -n "[$1]  "
  BENCH_FORCE_MOVER=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k[g] for g in ('A.fused','A.join_retune')})"
}
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
for r in 1 2; do
  for n in prev nofast bf2; do
    cp tools/experiments/bin/libedgehip_$n.so rebvo_amd/lib/libedgehip.so
    ab $n
  done
done 2>&1 | tee $OUT/ab.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
