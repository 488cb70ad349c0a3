#!/bin/bash
# FordwardMatch's arbitration + exp(W) + NaN check inside the out-of-place rotate_keylines pass (k_rotate<OUT, WIN>) against the previous
# library (k_fwd_win + k_rotate<OUT>): bit-identity test, frame-path suites, then A/B twice, same box.
set -u
OUT=$PWD/gpurun_out/r04_z; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_pipeline_gpu.py tests/test_soak_gpu.py tests/test_small_batch_gpu.py tests/test_edge_cases_gpu.py tests/test_host_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest3.txt
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
ab() {
  echo -n "[$1]  "
  BENCH_FORCE_MOVER=0 timeout 300 python bench.py --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('C.rotate','C.directed_matching','C.forward_match')})"
}
for r in 1 2; do
  cp tools/experiments/bin/libedgehip_prev.so rebvo_amd/lib/libedgehip.so; ab win_kernel
  cp /tmp/keep.so rebvo_amd/lib/libedgehip.so; ab win_in_rotate
done 2>&1 | tee $OUT/ab_win.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
