#!/bin/bash
OUT=$PWD/gpurun_out/r05_m; mkdir -p $OUT
timeout 600 python -m pytest tests/test_batch_group_gpu.py tests/test_host_gpu.py tests/test_dataset_gpu.py tests/test_imu_gpu.py tests/test_stereo_gpu.py -x -q > $OUT/pytest.log 2>&1
echo "tests exit $?"; grep -v "^REBVO\|^Advancing\|^Camara\|^Loaded\|^$" $OUT/pytest.log | tail -6
STAGES="bench" tools/gpu_round5.sh r05_m | cut -c1-200
python - <<'PY'
import json
js = json.load(open("gpurun_out/r05_m/bench_extras.json"))
hs = js.get("host_surface") or {}
print({k: v for k, v in hs.items() if k != "detail"})
PY
