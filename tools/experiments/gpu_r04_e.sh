#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_generality_gpu.py -q > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -25 $OUT/pytest.log
