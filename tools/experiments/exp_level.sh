#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_level_kernel_gpu.py tests/test_stage_a_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -4
EDGEHIP_LEVEL_MODE=2 python tools/prof_stage_a.py 256 2>&1 | grep -E "stage A|level|detect"
python bench.py --steps 20 --warmup 12 --cpu-frames 0
