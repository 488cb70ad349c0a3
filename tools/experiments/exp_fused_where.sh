#!/bin/bash
# Where a k_stage_a_fused workgroup's time goes: builds with -DEDGEHIP_FUSED_TSTAMP=k return an interval of the workgroup's life through kn_out
# (1 set-up, 2 ticks 0-9, 3 ticks 10-109, 4 ticks 110-end, 5 after the loop, 6 everything), 10 ns units.
#   for k in 1 2 3 4 5 6; do bash tools/experiments/build_variant.sh tstamp$k stage_a_fused.hip -DEDGEHIP_FUSED_TSTAMP=$k; done
cd "${GRAFT_REPO_ROOT:-.}"
cp rebvo_amd/lib/libedgehip.so /tmp/keep_ts.so
for k in 1 2 3 4 5 6; do
  cp tools/experiments/bin/libedgehip_tstamp$k.so rebvo_amd/lib/libedgehip.so
  EDGEHIP_LEVEL_MODE=3 python - $k <<'PY' 2>&1 | grep -v "^REBVO"
import sys
sys.path.insert(0, '.')
import numpy as np
from rebvo_amd import edgehip, synth
k = int(sys.argv[1]); B = 1024
frames = [f for f, _, _ in synth.billboard_sequence(752, 480, 3)]
eh = edgehip.EdgeHip(edgehip.euroc_params(752, 480), nseq=B, nslots=2)
for s in range(2): eh.upload_rgb(s, np.stack([frames[s]] * B))
for it in range(4): eh.stage_a(it % 2)
eh.sync()
v = np.array(eh.get_kn(1), dtype=np.float64) / 100.0
names = {1: "set-up", 2: "ticks 0-9", 3: "ticks 10-109", 4: "ticks 110-129", 5: "after the loop", 6: "whole workgroup"}
print(f"{names[k]:16s} mean {v.mean():8.1f} us  min {v.min():8.1f}  max {v.max():8.1f}")
PY
done
cp /tmp/keep_ts.so rebvo_amd/lib/libedgehip.so
