#!/bin/bash
# round 4, call q: build_field's rasteriser with the samples dealt to the threads in equal shares (EDGEHIP_RASTER_BAL=1, the new default)
# against one thread per half KeyLine (=0): field parity tests, then A/B through the bench, twice, same box.
set -u
OUT=$PWD/gpurun_out/r04_q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_stage_b_gpu.py tests/test_pipeline_gpu.py tests/test_generality_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt
ab() {
  echo -n "[$1]  "
  EDGEHIP_RASTER_BAL=$2 BENCH_FORCE_MOVER=0 timeout 300 python bench.py $3 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('B.build_field','B.try_velrot')})"
}
for r in 1 2; do
  ab per_keyline 0 ""
  ab balanced 1 ""
done 2>&1 | tee $OUT/ab.txt
ab per_keyline_tum 0 "--config tum_undistort" | tee -a $OUT/ab.txt
ab balanced_tum 1 "--config tum_undistort" | tee -a $OUT/ab.txt
