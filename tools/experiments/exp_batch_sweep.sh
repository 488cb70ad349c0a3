#!/bin/bash
# frames/s and per-frame latency of the full path against the number of sequences per launch
for B in 1 8 64 256 512 1024; do
  python bench.py --nseq $B --steps 30 --warmup 16 --cpu-frames 0 --no-roofline-events | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('nseq', $B, 'frames/s', d['value'], 'ms/step', d['ms_per_step'])"
done
echo "== torch.distributed.run, 1 rank, nccl"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 5 --cpu-frames 0 2>&1 | tail -2 | cut -c1-400
