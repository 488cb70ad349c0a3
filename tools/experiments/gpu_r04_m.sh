#!/bin/bash
# round 4, call m: the N > 1 branch of bench.py as two gloo ranks sharing the one GPU (control flow only), generality tests after the k_detect<WS> loop change
set -u
OUT=$PWD/gpurun_out/r04_m; mkdir -p $OUT
timeout 600 python -m pytest tests/test_generality_gpu.py -x -q 2>&1 | tail -3
BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 8 --warmup 4 --nseq 256 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks.err
echo "2-rank exit $?"; tail -c 600 $OUT/bench_2ranks.err
python - <<'PY'
import json
ls=[l for l in open("gpurun_out/r04_m/bench_2ranks_gloo.json").read().splitlines() if l.strip()]
print(len(ls), "line(s)")
j=json.loads(ls[0]); print(j["n_gpus"], j["value"], j["config"]["nav_gather"], j["config"]["nav_gather_info"], j["scaling"])
PY
