#!/bin/bash
OUT=$PWD/gpurun_out/r05_h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_batch_group_gpu.py tests/test_stage_b_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py tests/test_imu_gpu.py -x -q > $OUT/pytest_new.log 2>&1
echo "tests exit $?"; grep -v "^REBVO\|^Advancing\|^Camara" $OUT/pytest_new.log | tail -3
for r in 1 2; do timeout 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); k=json.load(open('bench_extras.json'))['kernel_us_per_step']; print(j['value'], j['ms_per_step'], {g: k.get(g) for g in ('A.fused','B.try_velrot','B.try_velrot2','B.build_field','C.directed_matching')})"; done
