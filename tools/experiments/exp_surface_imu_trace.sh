#!/bin/bash
# rocprofv3 kernel trace of bench.py's IMU-group surface leg (8 ImuMode=2 objects in one group): which kernels a step's 1.1 ms are made of
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, ".")
import bench
from rebvo_amd import edgehip, synth, config
W, H = 752, 480
p = edgehip.euroc_params(W, H)
intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 24, seed=11, **intr)]
np.stack(frames).tofile("/tmp/fi.rgb24")
bench._write_surface_imu_csv("/tmp/imu.csv", 24, 320, 1.0, 0.05)
config.write_global_config("/tmp/cfg_imu", p, imu=dict(mode=2, file="/tmp/imu.csv", time_scale=1.0, InitBiasFrameNum=3))
PY
OUT=$PWD/gpurun_out/imu_trace; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o t -- $GRAFT_REPO_ROOT/rebvo_amd/lib/surface_replay /tmp/cfg_imu /tmp/fi.rgb24 24 8 300 1 0.05 --warmup 40 --threads 8 --group g --stagger > $OUT/run.out 2> $OUT/run.err )
tail -1 $OUT/run.out | cut -c1-200
DB=$(ls $OUT/*.db $OUT/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py "$DB" "$OUT/kernel_stats.txt" > /dev/null
head -16 $OUT/kernel_stats.txt | cut -c1-150
[ -n "$DB" ] && python tools/rocpd_timeline.py "$DB" k_imu_filter 200 | cut -c1-110 > $OUT/timeline.txt
find $OUT -name "*.db" -delete
