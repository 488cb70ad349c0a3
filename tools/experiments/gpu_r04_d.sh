#!/bin/bash
# round 4, call d: generality (any width, plane-fit windows) + the stage-level tests it touches + the remaining tests of call c
set -u
OUT=$PWD/gpurun_out/r04_d; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_generality_gpu.py tests/test_stage_a_gpu.py tests/test_fused_stage_a_gpu.py tests/test_grey8_gpu.py tests/test_edge_cases_gpu.py tests/test_stereo_gpu.py tests/test_undistort_gpu.py tests/test_level_kernel_gpu.py -q > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -25 $OUT/pytest.log
