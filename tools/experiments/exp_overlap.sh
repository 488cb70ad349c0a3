#!/bin/bash
for O in 0 1; do
echo "== EDGEHIP_OVERLAP=$O"
EDGEHIP_OVERLAP=$O timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -1
EDGEHIP_OVERLAP=$O python bench.py --steps 20 --warmup 12 --cpu-frames 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
done
