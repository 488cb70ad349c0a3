#!/bin/bash
# round 4, call p: review item 3 measured — the undistortion inside the one-kernel stage A's load (EDGEHIP_FUSED_UNDIST=1, SRC_UNDIST)
# against the pre-pass (k_undistort_grey + the 16-bit plane): parity tests, then the TUM configuration both ways, twice, same box.
set -u
OUT=$PWD/gpurun_out/r04_p; mkdir -p $OUT
timeout 900 python -m pytest tests/test_fused_stage_a_gpu.py tests/test_undistort_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
ab() {
  echo -n "[$1]  "
  EDGEHIP_FUSED_UNDIST=$2 BENCH_FORCE_MOVER=0 timeout 300 python bench.py --config tum_undistort --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('A.fused','A.rgb_rowscan','A.join_retune')})"
}
for r in 1 2; do
  ab prepass 0
  ab in_load 1
done 2>&1 | tee $OUT/ab.txt
