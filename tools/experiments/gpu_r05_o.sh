#!/bin/bash
OUT=$PWD/gpurun_out/r05_o; mkdir -p $OUT
timeout 300 python -m pytest tests/test_fused_stage_a_gpu.py tests/test_golden_gpu.py -x -q > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -2 $OUT/pytest.log
timeout 600 tools/experiments/ab_libs.sh base add16 2>&1 | tee $OUT/scan_add16_ab.txt
