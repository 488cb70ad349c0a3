#!/bin/bash
# From how many sequences per launch on is the one-kernel stage A (one workgroup per sequence) the better choice?  (EDGEHIP_FUSED_MIN_BATCH, default 192)
cd "${GRAFT_REPO_ROOT:-.}"
for n in 24 32 48 64 96 128 160; do
  for mb in 192 16; do
    echo -n "nseq $n  min_batch $mb  "
    EDGEHIP_FUSED_MIN_BATCH=$mb timeout 300 python bench.py --nseq $n --steps 200 --warmup 30 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys, json
l = sys.stdin.read(); j = json.loads(l[l.index('{'):]); print(j['value'], j['ms_per_step'])"
  done
done
