#!/bin/bash
OUT=$PWD/gpurun_out/r05_f; mkdir -p $OUT
timeout 500 python -m pytest tests/test_batch_group_gpu.py tests/test_host_gpu.py tests/test_shard_gpu.py tests/test_edge_cases_gpu.py tests/test_pipeline_gpu.py -x -q > $OUT/pytest_new.log 2>&1
echo "tests exit $?"; grep -v "^REBVO" $OUT/pytest_new.log | tail -3
STAGES="bench" tools/gpu_round5.sh r05_f | cut -c1-300
python - <<'PY'
import json
js = json.load(open("gpurun_out/r05_f/bench_extras.json"))
hs = js.get("host_surface") or {}
print({k: v for k, v in hs.items() if k != "detail"})
print({k: v.get("ms_per_step") for k, v in (hs.get("detail") or {}).items()})
PY
