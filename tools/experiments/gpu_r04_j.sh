#!/bin/bash
# round 4, call j: the reweighted evaluation with two KeyLines per thread: parity, then A/B (threshold env: 0 = off)
set -u
OUT=$PWD/gpurun_out/r04_j; mkdir -p $OUT
timeout 600 python -m pytest tests/test_pipeline_gpu.py -x -q -k "two_keylines or two_chain" 2>&1 | tail -4
EDGEHIP_TVR_RW2=1 timeout 900 python -m pytest tests/test_soak_gpu.py tests/test_stage_b_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -4
ab() {
  echo -n "[$1]  "
  env $1 BENCH_FORCE_MOVER=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k[g] for g in ('B.try_velrot','B.lm_step')})"
}
for r in 1 2 3; do
  ab EDGEHIP_TVR_RW2=0
  ab EDGEHIP_TVR_RW2=4096
done 2>&1 | tee $OUT/ab.txt
