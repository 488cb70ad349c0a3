#!/bin/bash
# Frames per second against the sequences per launch, with and without stage A of the next frame beside this frame's tracking (EDGEHIP_OVERLAP)
cd "${GRAFT_REPO_ROOT:-.}"
for n in ${NSEQS:-256 512 768 1024}; do
  for o in 0 1; do
    echo -n "nseq $n  overlap $o  "
    EDGEHIP_OVERLAP=$o timeout 300 python bench.py --nseq $n --steps ${STEPS:-40} --warmup 10 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys, json
l = sys.stdin.read(); j = json.loads(l[l.index('{'):]); k = json.load(open('bench_extras.json'))['kernel_us_per_step']
print(j['value'], j['ms_per_step'], 'kernel sum', round(sum(k.values())), 'A.fused', round(k.get('A.fused', 0)), 'roofline launch_us', j['roofline'].get('launch_us'))"
  done
done
