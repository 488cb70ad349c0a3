#!/bin/bash
for ARGS in "--contexts 1 --nseq 256" "--contexts 2 --nseq 512" "--contexts 3 --nseq 384"; do
python bench.py --steps 20 --warmup 12 --cpu-frames 0 $ARGS | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$ARGS', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['config']['estimation_ok'])"
done
