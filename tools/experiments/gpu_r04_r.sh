#!/bin/bash
# round 4, call r: where does k_field_raster's time go?  Ablations of the per-KeyLine rasteriser (wrong fields by design, timing only):
# 1 = plain LDS store instead of the atomic, 2 = no LDS access, 3 = set-up only (no samples).  B.build_field includes k_field_bin (~190 us).
set -u
OUT=$PWD/gpurun_out/r04_r; mkdir -p $OUT
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
ab() {
  echo -n "[$1]  "
  EDGEHIP_RASTER_BAL=0 BENCH_FORCE_MOVER=0 timeout 300 python bench.py --no-extras --cpu-frames 0 --steps 10 --warmup 4 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('B.build_field',)}, j['config'].get('keylines_per_frame_timed_mean'))"
}
ab full
for v in 3 4 5 6; do
  cp tools/experiments/bin/libedgehip_rabl$v.so rebvo_amd/lib/libedgehip.so
  ab abl$v
done 2>&1 | tee $OUT/ab.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
