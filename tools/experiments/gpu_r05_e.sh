#!/bin/bash
OUT=$PWD/gpurun_out/r05_e; mkdir -p $OUT
timeout 400 python -m pytest tests/test_batch_group_gpu.py tests/test_host_gpu.py tests/test_pipeline_gpu.py tests/test_soak_gpu.py tests/test_fused_stage_a_gpu.py tests/test_stage_b_gpu.py -x -q > $OUT/pytest_new.log 2>&1
echo "tests exit $?"; grep -v "^REBVO" $OUT/pytest_new.log | tail -4
STAGES="bench" tools/gpu_round5.sh r05_e | cut -c1-600
python - <<'PY'
import json
js = json.load(open("gpurun_out/r05_e/bench_extras.json"))
print(json.dumps(js.get("host_surface"), indent=1)[:3000])
PY
