#!/bin/bash
# round 4, call g: rasteriser (two samples per trip, per-tile split) and fit-wave (no window copy) variants: parity, then same-box A/B
set -u
OUT=$PWD/gpurun_out/r04_g; mkdir -p $OUT
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
cp tools/experiments/bin/libedgehip_fitun.so rebvo_amd/lib/libedgehip.so
timeout 600 python -m pytest tests/test_fused_stage_a_gpu.py -x -q 2>&1 | tail -3
cp tools/experiments/bin/libedgehip_ras2adapt.so rebvo_amd/lib/libedgehip.so
timeout 600 python -m pytest tests/test_stage_b_gpu.py -x -q -k "field" 2>&1 | tail -3
ab() {
  echo -n "[$1]  "
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k[g] for g in ('A.fused','B.build_field','B.try_velrot')})"
}
for r in 1 2; do
  for n in base fitun ras2 adapt ras2adapt; do
    cp tools/experiments/bin/libedgehip_$n.so rebvo_amd/lib/libedgehip.so
    BENCH_FORCE_MOVER=0 ab $n
  done
done 2>&1 | tee $OUT/ab.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
