#!/bin/bash
# round 5, trip c: the new host tests first (short), fused-kernel ablations, then the whole suite and the bench
OUT=$PWD/gpurun_out/r05_c; mkdir -p $OUT
timeout 400 python -m pytest tests/test_batch_group_gpu.py tests/test_host_gpu.py tests/test_golden_gpu.py -x -q > $OUT/pytest_new.log 2>&1
echo "new tests exit $?"; tail -5 $OUT/pytest_new.log
VARIANTS="0 1 8 9 62 2 16 32" timeout 400 tools/experiments/exp_fused_ablate.sh 1024 > $OUT/fused_ablate.txt 2>&1
cat $OUT/fused_ablate.txt
if grep -q "passed" $OUT/pytest_new.log && ! grep -q "failed" $OUT/pytest_new.log; then
  STAGES="bench tests" tools/gpu_round5.sh r05_c
fi
