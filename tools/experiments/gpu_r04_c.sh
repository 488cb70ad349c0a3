#!/bin/bash
# round 4, call c: the whole GPU suite after the ADVICE fixes + the driver-form bench line
set -u
OUT=$PWD/gpurun_out/r04_c; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; wc -l $OUT/bench.json; tail -c 600 $OUT/bench.err
python - <<'PY'
import json
l = open("gpurun_out/r04_c/bench.json").read()
j = json.loads(l)
print(j["value"], j["ms_per_step"], j["config"]["nav_gather"], j["config"]["nav_gather_info"])
print(j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline_kernels"]["B.try_velrot"])
print(j["kernel_us_per_step"])
print(j["pose_rmse"]["free_running_parity"])
PY
