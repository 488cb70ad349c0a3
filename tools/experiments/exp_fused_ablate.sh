#!/bin/bash
# fused stage-A kernel: phase ablations (needs a `make EXPERIMENTS=1` build).  bits: 1 scan, 2 box chain, 4 window tests,
# 8 KeyLine emit + mask rows, 16 RGB loads, 32 gradient gate
cd "$GRAFT_REPO_ROOT"
B=${1:-1024}
for A in 0 1 2 4 8 16 32 63 62; do
  echo "== EDGEHIP_FUSED_ABLATE=$A"
  EDGEHIP_FUSED_ABLATE=$A python tools/prof_stage_a.py $B 2>&1 | grep -E "stage A|fused"
done
