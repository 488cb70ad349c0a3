#!/bin/bash
# round 4, call h: contexts / overlap A/B at 1024 sequences with the round's kernels
set -u
OUT=$PWD/gpurun_out/r04_h; mkdir -p $OUT
ab() {
  echo -n "[$*]  "
  BENCH_FORCE_MOVER=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 "$@" 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); print(j['value'], j['ms_per_step'])"
}
for r in 1 2; do
  ab
  ab --contexts 2
  ab --contexts 4
  ab --overlap
  ab --overlap --contexts 2
  ab --nseq 2048
  ab --nseq 2048 --contexts 2
done 2>&1 | tee $OUT/ab.txt
