#!/bin/bash
# The other configurations at mid-size batches (the one-kernel stage A beside the previous frame's tracking since the threshold is 32): each line's pose check
# against the reference (33 or fewer sequences) must come out with nothing outside tolerance.
cd "${GRAFT_REPO_ROOT:-.}"
run() { echo -n "$* :  "; timeout 600 python bench.py "$@" --steps 30 --warmup 10 --no-extras --cpu-frames 20 2>/dev/null | python -c "
import sys, json
l = sys.stdin.read(); j = json.loads(l[l.index('{'):]); p = j.get('pose_rmse') or {}
print(j['value'], j['ms_per_step'], 'est ok', j['config'].get('estimation_ok'), 'checked', p.get('sequences_checked'), 'outside', p.get('outside_tolerance'), 'elsewhere', p.get('departures_elsewhere'))"; }
for n in 48 96; do
  run --nseq $n
  run --nseq $n --imu
  run --nseq $n --config tum_undistort
  run --nseq $n --tracker-f32
done
