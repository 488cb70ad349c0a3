#!/bin/bash
# A/B of environment switches on one box: ab_env.sh "VAR=val" "VAR2=val" ...   ("" = default); two rounds each
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for m in "$@"; do
  echo -n "[$m]  "
  env $m timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k[g] for g in k if g.startswith('B.')})"
done
done
