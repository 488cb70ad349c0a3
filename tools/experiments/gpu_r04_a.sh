#!/bin/bash
# round 4, call a: the two-chain initialisation — parity first, then same-box A/B of its variants against the launch chain
set -u
OUT=$PWD/gpurun_out/r04_a; mkdir -p $OUT
timeout 600 python -m pytest tests/test_pipeline_gpu.py tests/test_stage_b_gpu.py tests/test_soak_gpu.py tests/test_small_batch_gpu.py -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/pytest.log
ab() {  # name env
  echo -n "[$1 $2]  "
  env $2 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k[g] for g in k if g.startswith('B.')})"
}
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so
for r in 1 2; do
  for n in park nopark park8; do
    cp tools/experiments/bin/libedgehip_$n.so rebvo_amd/lib/libedgehip.so
    ab $n EDGEHIP_DUAL_INIT=1
  done
  cp tools/experiments/bin/libedgehip_park.so rebvo_amd/lib/libedgehip.so
  ab chain EDGEHIP_DUAL_INIT=0
done 2>&1 | tee $OUT/ab.txt
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
# single camera / small batches
for m in EDGEHIP_DUAL_INIT=0 EDGEHIP_DUAL_INIT=1; do
  for n in 1 8 64; do
    echo -n "[$m nseq $n] "
    env $m timeout 200 python bench.py --nseq $n --steps 200 --warmup 30 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); print(j['value'], j['ms_per_step'])"
  done
done 2>&1 | tee $OUT/small.txt
