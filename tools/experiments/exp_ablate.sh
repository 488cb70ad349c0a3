#!/bin/bash
# detect-kernel ablations (stage A only, 256 sequences)
for A in 0 2 4 6; do
  echo "== EDGEHIP_ABLATE=$A"
  EDGEHIP_ABLATE=$A python tools/prof_stage_a.py 256 2>&1 | grep -E "stage A|detect|avg|colscan|rowscan"
done
