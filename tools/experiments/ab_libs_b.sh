#!/bin/bash
# A/B of whole libraries on one box, stage-B / C view: ab_libs_b.sh name1 name2 ...  (tools/experiments/bin/libedgehip_<name>.so), two rounds
cd $GRAFT_REPO_ROOT
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
for r in 1 2; do
for n in "$@"; do
  cp tools/experiments/bin/libedgehip_$n.so rebvo_amd/lib/libedgehip.so
  echo -n "[$n]  "
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k[g] for g in ('B.build_field','B.try_velrot','C.directed_matching','C.regularize_ekf','C.rescale')})"
done
done
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
