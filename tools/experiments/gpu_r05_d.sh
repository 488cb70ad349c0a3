#!/bin/bash
OUT=$PWD/gpurun_out/r05_d; mkdir -p $OUT
timeout 400 python -m pytest tests/test_batch_group_gpu.py tests/test_host_gpu.py tests/test_golden_gpu.py -x -q > $OUT/pytest_new.log 2>&1
echo "new tests exit $?"; grep -v "^REBVO" $OUT/pytest_new.log | tail -5
timeout 300 tools/experiments/exp_fused_occupancy.sh > $OUT/fused_occupancy.txt 2>&1; cat $OUT/fused_occupancy.txt
if grep -q "passed" $OUT/pytest_new.log && ! grep -q "failed" $OUT/pytest_new.log; then
  STAGES="bench tests" tools/gpu_round5.sh r05_d
fi
