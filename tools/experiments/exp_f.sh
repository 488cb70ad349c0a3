#!/bin/bash
for A in 0 1 2; do
EDGEHIP_FIELD_ABLATE=$A python bench.py --steps 10 --warmup 12 --cpu-frames 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ablate', $A, d['kernel_us_per_step']['B.build_field'])"
done
