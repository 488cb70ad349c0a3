#!/bin/bash
# Where a 1024-member group spends its time: surface_replay under REBVO_GROUP_TIMING=3 (progress lines on stderr), bounded by `timeout`.
N=${1:-1024}; K=${2:-30}; T=${3:-16}; LIM=${4:-100}
python - <<PY
import numpy as np, sys
sys.path.insert(0, "/root/repo")
from rebvo_amd import edgehip, synth, config
W, H = 752, 480
p = edgehip.euroc_params(W, H)
intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 8, seed=11, **intr)]
np.stack(frames).tofile("/tmp/f8.rgb24")
config.write_global_config("/tmp/cfg_s", p)
PY
date +%s.%N
SURFACE_REPLAY_PHASES=1 REBVO_GROUP_TIMING=1 timeout $LIM rebvo_amd/lib/surface_replay /tmp/cfg_s /tmp/f8.rgb24 8 $N $K 1 0.05 --warmup 10 --threads $T --group big > /tmp/sr.out 2> /tmp/sr.err
echo "exit $?"; date +%s.%N
tail -3 /tmp/sr.out | cut -c1-600
grep -c "step" /tmp/sr.err; head -4 /tmp/sr.err; sed -n '5,12p' /tmp/sr.err; tail -4 /tmp/sr.err
