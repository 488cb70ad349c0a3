#!/bin/bash
OUT=$PWD/gpurun_out/r05_n; mkdir -p $OUT
fail=0
for i in $(seq 1 12); do
  timeout 200 python -m pytest tests/test_batch_group_gpu.py tests/test_host_gpu.py tests/test_dataset_gpu.py -x -q > $OUT/loop_$i.log 2>&1 || { fail=$((fail+1)); echo "iteration $i FAILED"; grep -v "^REBVO\|^Advancing\|^Camara\|^Loaded\|^$" $OUT/loop_$i.log | tail -15; }
done
echo "failures: $fail of 12"
