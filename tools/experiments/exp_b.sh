#!/bin/bash
timeout 800 python -m pytest tests/test_stage_b_gpu.py tests/test_pipeline_gpu.py tests/test_golden_gpu.py tests/test_undistort_gpu.py -x -q 2>&1 | tail -4
python bench.py --steps 20 --warmup 12 --cpu-frames 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step']); print(d['kernel_us_per_step'])"
