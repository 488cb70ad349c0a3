#!/bin/bash
OUT=$PWD/gpurun_out/r05_g; mkdir -p $OUT
timeout 600 python -m pytest tests/test_stage_c_gpu.py tests/test_golden_gpu.py tests/test_pipeline_gpu.py tests/test_edge_cases_gpu.py tests/test_soak_gpu.py -x -q > $OUT/pytest_dm.log 2>&1
echo "tests exit $?"; grep -v "^REBVO" $OUT/pytest_dm.log | tail -3
timeout 600 tools/experiments/ab_libs.sh dm_p0w7 dm_p2w7 dm_p3w7 dm_p2w6 dm_p3w6 dm_p5w6 2>&1 | tee $OUT/directed_two_phase_ab.txt
