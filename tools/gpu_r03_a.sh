#!/bin/bash
# round 3, call A: knife-edge parity tests + evidence, and the 2-ranks-on-1-GPU dry run of bench.py's N > 1 branch
set -u
OUT=$PWD/gpurun_out/r03_a
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_knife_edge_gpu.py tests/test_shard_gloo.py -x -q > "$OUT/pytest_knife_edge.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_knife_edge.log"; tail -15 "$OUT/pytest_knife_edge.log"
timeout 300 python tools/experiments/exp_knife_edge_cpu.py 10 > "$OUT/knife_edge_cpu_epyc.txt" 2>&1; echo "cpu exp exit $?"
head -12 "$OUT/knife_edge_cpu_epyc.txt" | cut -c1-200
timeout 300 python tools/experiments/exp_knife_edge_gpu.py 5 14 > "$OUT/knife_edge_gpu_seq5.json" 2> "$OUT/knife_edge_gpu.err"; echo "gpu exp exit $?"
head -c 1500 "$OUT/knife_edge_gpu_seq5.json"; tail -3 "$OUT/knife_edge_gpu.err"
BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
   bench.py --gpus 2 --nseq 256 --steps 8 --warmup 6 --cpu-frames 0 > "$OUT/bench_2ranks_gloo.json" 2> "$OUT/bench_2ranks_gloo.err"
echo "2-rank exit $?"; tail -c 1500 "$OUT/bench_2ranks_gloo.json"; tail -5 "$OUT/bench_2ranks_gloo.err"
