#!/bin/bash
# on the GPU box: time the fused stage-A kernel of every library tools/experiments/bin/libedgehip_<name>.so given
cd "$GRAFT_REPO_ROOT"
B=${B:-1024}
cp rebvo_amd/lib/libedgehip.so /tmp/libedgehip_keep.so
echo -n "default  "; EDGEHIP_LEVEL_MODE=3 python tools/prof_stage_a.py $B 2>&1 | grep -E "fused" | awk '{print $2, $3}'
for A in "$@"; do
  cp tools/experiments/bin/libedgehip_$A.so rebvo_amd/lib/libedgehip.so
  echo -n "$A  "
  EDGEHIP_LEVEL_MODE=3 python tools/prof_stage_a.py $B 2>&1 | grep -E "fused" | awk '{print $2, $3}'
done
cp /tmp/libedgehip_keep.so rebvo_amd/lib/libedgehip.so
