#!/bin/bash
# round 4, call b: the new tests (RCCL world-1 mover, multi-device replay, 752x480 / >= 192-sequence IMU, stereo, key-frame) and the default bench line
set -u
OUT=$PWD/gpurun_out/r04_b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_shard_gpu.py tests/test_dataset_gpu.py tests/test_stage_b_gpu.py tests/test_stereo_gpu.py tests/test_imu_gpu.py -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3
echo "bench exit $?"; tail -c 1500 $OUT/bench.err
python - <<'PY'
import json
l = open("gpurun_out/r04_b/bench.json").read()
j = json.loads(l[l.index("{"):])
print(j["value"], j["ms_per_step"], j["config"]["nav_gather"], j["config"]["nav_gather_info"])
print(j["roofline"])
print({k: v for k, v in j["pose_rmse"].items() if k not in ("position_per_sequence",)})
print(j.get("single_sequence_ms_per_frame"), j.get("batch_sweep"))
h = j.get("heterogeneous", {})
print({k: h.get(k) for k in ("value", "free_running_parity")})
print(j["kernel_us_per_step"])
PY
