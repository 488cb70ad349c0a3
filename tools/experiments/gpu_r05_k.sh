#!/bin/bash
OUT=$PWD/gpurun_out/r05_k; mkdir -p $OUT
for r in 1 2 3; do for m in 0 1; do echo -n "[EDGEHIP_GRAPH=$m] "; EDGEHIP_GRAPH=$m timeout 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['launch_us'])"; done; done 2>&1 | tee $OUT/graph_ab.txt
