#!/bin/bash
OUT=$PWD/gpurun_out/r05_j; mkdir -p $OUT
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_stage_b_gpu.py tests/test_golden_gpu.py tests/test_small_batch_gpu.py tests/test_knife_edge_gpu.py -x -q > $OUT/pytest_compact.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest_compact.log
for r in 1 2 3; do for m in 0 192; do echo -n "[compact_min=$m] "; EDGEHIP_TVR_COMPACT_MIN=$m timeout 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); k=json.load(open('bench_extras.json'))['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('B.try_velrot','B.try_velrot2','B.lm_step','B.tvr_prepare','B.quantile')})"; done; done 2>&1 | tee $OUT/tvr_compact_ab.txt
