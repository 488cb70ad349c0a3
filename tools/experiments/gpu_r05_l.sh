#!/bin/bash
OUT=$PWD/gpurun_out/r05_l; mkdir -p $OUT
timeout 600 python -m pytest tests/test_dataset_gpu.py tests/test_batch_group_gpu.py tests/test_host_gpu.py -x -q > $OUT/pytest.log 2>&1
echo "tests exit $?"; grep -v "^REBVO\|^Advancing\|^Camara\|^Loaded\|^$" $OUT/pytest.log | tail -12
