#!/bin/bash
# What the pipeline fill / drain of a k_stage_a_fused workgroup costs: the same 1024-sequence launch at three image heights (the box chain's
# latency in ticks does not depend on the height).  T(h) = fixed + per_row * h  ->  fixed = the part a frame streamed behind another would not pay.
cd "${GRAFT_REPO_ROOT:-.}"
for H in 240 480 720 960; do
  echo -n "h=$H  "
  EDGEHIP_LEVEL_MODE=3 python tools/prof_stage_a.py 1024 752 $H 2>&1 | grep -E "fused" | awk '{print $2, $3}'
done
