#!/bin/bash
OUT=$PWD/gpurun_out/r05_i; mkdir -p $OUT
timeout 600 python -m pytest tests/test_fused_stage_a_gpu.py tests/test_golden_gpu.py tests/test_grey8_gpu.py tests/test_generality_gpu.py -x -q > $OUT/pytest_scan.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest_scan.log
timeout 600 tools/experiments/ab_libs.sh scan_old scan_dpp 2>&1 | tee $OUT/scan_dpp_ab.txt
