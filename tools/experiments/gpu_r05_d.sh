#!/bin/bash
OUT=$PWD/gpurun_out/r05_d; mkdir -p $OUT
timeout 400 python -m pytest tests/test_batch_group_gpu.py tests/test_host_gpu.py tests/test_golden_gpu.py -x -q > $OUT/pytest_new.log 2>&1
echo "new tests exit $?"; grep -v "^REBVO" $OUT/pytest_new.log | tail -5
timeout 300 tools/experiments/exp_fused_occupancy.sh > $OUT/fused_occupancy.txt 2>&1; cat $OUT/fused_occupancy.txt
# stream overlap / two contexts on the current kernels (the default line runs one context, no overlap)
for V in "" "--overlap" "--contexts 2" "--contexts 2 --overlap"; do
  echo -n "bench $V : "
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $V 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['launch_us'])"
done 2>&1 | tee $OUT/overlap_contexts.txt
# two ranks sharing the one GPU over gloo: the N > 1 control flow of bench.py end to end
BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
   bench.py --gpus 2 --steps 8 --warmup 4 --nseq 64 --cpu-frames 30 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo "2-rank gloo exit $?"; tail -c 1500 $OUT/bench_2rank_gloo.json
if grep -q "passed" $OUT/pytest_new.log && ! grep -q "failed" $OUT/pytest_new.log; then
  STAGES="bench tests" tools/gpu_round5.sh r05_d
fi
