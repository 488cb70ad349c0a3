#!/bin/bash
# A/B of whole libraries on one box: ab_libs.sh name1 name2 ...  (tools/experiments/bin/libedgehip_<name>.so), three rounds
cd $GRAFT_REPO_ROOT
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
for r in 1 2 3; do
for n in "$@"; do
  cp tools/experiments/bin/libedgehip_$n.so rebvo_amd/lib/libedgehip.so
  echo -n "[$n]  "
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=json.load(open('bench_extras.json'))['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('A.fused','B.try_velrot','B.build_field','C.rotate','C.directed_matching')})"
done
done
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
