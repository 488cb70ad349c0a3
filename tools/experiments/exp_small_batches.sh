#!/bin/bash
# Small batches: per-group kernel time against the step, at 1 / 8 / 64 sequences per launch (what the plugin surface's single camera and small groups run)
cd "${GRAFT_REPO_ROOT:-.}"
for n in ${NSEQS:-1 8 64 192 256}; do
  timeout 300 python bench.py --nseq $n --steps 200 --warmup 30 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys, json
l = sys.stdin.read(); j = json.loads(l[l.index('{'):]); k = json.load(open('bench_extras.json'))['kernel_us_per_step']
tot = sum(k.values())
print('nseq', $n, 'fps', j['value'], 'ms/step', j['ms_per_step'], 'kernels sum us', round(tot, 1), {a: round(b, 1) for a, b in sorted(k.items(), key=lambda x: -x[1])})"
done
