#!/bin/bash
timeout 800 python -m pytest tests/test_stage_a_gpu.py tests/test_level_kernel_gpu.py tests/test_undistort_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -4
for A in 0 2 6; do echo "ablate $A"; EDGEHIP_ABLATE=$A python tools/prof_stage_a.py 256 2>&1 | grep -E "stage A|detect"; done
