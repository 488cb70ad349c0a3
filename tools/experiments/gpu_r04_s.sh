#!/bin/bash
# round 4, call s: one TryVelRot evaluation against the reference's, reference-order roundings (default) vs the single scale factor; and the
# generality tests again (the balanced rasteriser of call q had failed one of them; it is gone).
set -u
OUT=$PWD/gpurun_out/r04_s; mkdir -p $OUT
timeout 600 python -m pytest tests/test_generality_gpu.py tests/test_stage_b_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.txt
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
for n in reforder fastscale; do
  [ $n = fastscale ] && cp tools/experiments/bin/libedgehip_fastscale.so rebvo_amd/lib/libedgehip.so
  echo "== $n"; timeout 300 python tools/experiments/exp_tvr_closeness.py 752 480 2>&1 | tail -11
done | tee $OUT/closeness.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
