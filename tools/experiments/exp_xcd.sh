#!/bin/bash
for X in 0 1; do
EDGEHIP_XCD=$X python bench.py --steps 20 --warmup 12 --cpu-frames 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernel_us_per_step']
print('xcd', $X, d['value'], 'tvr', k['B.try_velrot'], 'detect', k['A.detect'], 'directed', k['C.directed_matching'], 'field', k['B.build_field'])"
done
