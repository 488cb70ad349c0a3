#!/bin/bash
# The scan wave's dependent adds as plain C instead of one asm statement each (the hazard recogniser puts an s_nop 0 behind every
# inline-asm VALU write; plain adds get none): micro-benchmark, parity of the variant, same-box A/B.
#   bash tools/experiments/build_variant.sh scanasm1 stage_a_fused.hip -DEDGEHIP_SCAN_ASM=1
#   bash tools/experiments/build_variant.sh scanasm0 stage_a_fused.hip -DEDGEHIP_SCAN_ASM=0
#   gpurun -- 'bash tools/experiments/exp_scan_plain_adds.sh'
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r06_scan_plain; mkdir -p $OUT
tools/experiments/bin/ubench_scan2 > $OUT/ubench.txt 2>&1; cat $OUT/ubench.txt
cp rebvo_amd/lib/libedgehip.so /tmp/keep0.so
cp tools/experiments/bin/libedgehip_scanasm0.so rebvo_amd/lib/libedgehip.so
timeout 900 python -m pytest tests/test_fused_stage_a_gpu.py tests/test_pipeline_gpu.py tests/test_golden_gpu.py -x -q > $OUT/pytest_scanasm0.log 2>&1; echo "variant tests exit $?"; grep -v "^REBVO" $OUT/pytest_scanasm0.log | tail -3
cp /tmp/keep0.so rebvo_amd/lib/libedgehip.so
bash tools/experiments/gpu_call.sh r06_scan_plain --ab-libs "scanasm1 scanasm0" --groups "A.fused A.join_retune"
