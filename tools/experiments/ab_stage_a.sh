#!/bin/bash
# A/B two builds of libedgehip.so on the SAME box (boxes differ by 10-15 %), stage A only: tools/experiments/bin/libedgehip_{A,B}.so
cd "$GRAFT_REPO_ROOT"
cp rebvo_amd/lib/libedgehip.so /tmp/libedgehip_keep.so
for v in A B A B A B; do
  cp tools/experiments/bin/libedgehip_$v.so rebvo_amd/lib/libedgehip.so
  echo -n "$v  "
  EDGEHIP_LEVEL_MODE=${EDGEHIP_LEVEL_MODE:-3} python tools/prof_stage_a.py ${1:-1024} 2>&1 | grep -E "fused|level|detect|compact" | awk '{printf "%s %s  ", $1, $2} END {print ""}'
done
cp /tmp/libedgehip_keep.so rebvo_amd/lib/libedgehip.so
