#!/bin/bash
# round 4, call l: the driver's form of the default line with the round's final kernels
set -u
OUT=$PWD/gpurun_out/r04_l; mkdir -p $OUT
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04_l/bench_driver_form.json").read())
print(j["value"], j["ms_per_step"], j["config"]["nav_gather"], j["single_sequence_ms_per_frame"])
print(json.dumps(j["roofline"])[:900])
print(j["kernel_us_per_step"])
print({k: (v["frac"], v.get("frac_on_traffic")) for k, v in j["roofline_kernels"].items()})
print(j["pose_rmse"]["free_running_parity"]["sequences_outside_tolerance_at_last_frame"], j["heterogeneous"]["free_running_parity"]["sequences_outside_tolerance_at_last_frame"])
PY
