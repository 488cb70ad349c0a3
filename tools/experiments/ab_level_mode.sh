#!/bin/bash
# stage A on the SAME box: multi-kernel path (EDGEHIP_LEVEL_MODE=2) against the fused kernel (3)
cd "$GRAFT_REPO_ROOT"
for m in 2 3 2 3; do
  echo -n "mode $m  "
  EDGEHIP_LEVEL_MODE=$m python tools/prof_stage_a.py ${1:-1024} 2>&1 | grep -E "stage A|fused|level|detect|compact|join" | awk '{printf "%s %s  ", $1, $2} END {print ""}'
done
