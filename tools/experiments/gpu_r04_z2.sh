#!/bin/bash
# k_directed<FUSED, FILL> at 74 registers (6 waves per SIMD) against the same kernel compiled for 7 waves (70 registers, 2 KB of LDS for its small arrays, 16 B scratch)
set -u
OUT=$PWD/gpurun_out/r04_z; mkdir -p $OUT
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
ab() {
  echo -n "[$1]  "
  BENCH_FORCE_MOVER=0 timeout 300 python bench.py --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('C.rotate','C.directed_matching','C.forward_match')})"
}
for r in 1 2; do
  cp /tmp/keep.so rebvo_amd/lib/libedgehip.so; ab occ6
  cp tools/experiments/bin/libedgehip_dir7.so rebvo_amd/lib/libedgehip.so; ab occ7
done 2>&1 | tee $OUT/ab_occ.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
