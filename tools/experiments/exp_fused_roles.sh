#!/bin/bash
# Who waits for whom inside a tick of k_stage_a_fused: per role, cycles per tick (mean over ticks 10..109 and over 1024 workgroups) of work before the
# middle barrier, wait at it, work before the end barrier, wait at it.  Builds: -DEDGEHIP_FUSED_TSTAMP=RX (R: 1 / 4 / 5 column waves 0 / 2 / 5, 2 scan, 3 fit).
cd "${GRAFT_REPO_ROOT:-.}"
cp rebvo_amd/lib/libedgehip.so /tmp/keep_ts.so
for r in 1 4 5 2 3; do
  for x in 1 2 3 4; do
  cp tools/experiments/bin/libedgehip_tstamp$r$x.so rebvo_amd/lib/libedgehip.so
  EDGEHIP_LEVEL_MODE=3 python - $r $x <<'PY' 2>&1 | grep -v "^REBVO"
import sys
sys.path.insert(0, '.')
import numpy as np
from rebvo_amd import edgehip, synth
r, x = int(sys.argv[1]), int(sys.argv[2]); B = 1024
frames = list(synth.rects_sequence(752, 480, 3))
eh = edgehip.EdgeHip(edgehip.euroc_params(752, 480), nseq=B, nslots=2)
for s in range(2): eh.upload_rgb(s, np.stack([frames[s]] * B))
for it in range(14): eh.stage_a(it % 2)
eh.sync()
v = np.array(eh.get_kn(1), dtype=np.float64)
role = {1: "column wave 0 (SIMD 0)", 4: "column wave 2 (SIMD 3, with fit)", 5: "column wave 5 (SIMD 2, with scan)", 2: "scan wave", 3: "fit wave"}[r]
what = {1: "work 1", 2: "wait mid", 3: "work 2", 4: "wait end"}[x]
print(f"{role:34s} {what:9s} {v.mean():8.0f} cycles/tick  (min {v.min():6.0f} max {v.max():6.0f})")
PY
  done
done
cp /tmp/keep_ts.so rebvo_amd/lib/libedgehip.so
