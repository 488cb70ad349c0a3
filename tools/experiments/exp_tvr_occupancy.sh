#!/bin/bash
# k_try_velrot against the number of resident blocks per CU (unused dynamic LDS caps them): is the kernel bound by latency
# (time grows as occupancy falls) or by the memory system's throughput (flat)?
cd $GRAFT_REPO_ROOT
cp rebvo_amd/lib/libedgehip.so /tmp/keep.so; cp tools/experiments/bin/libedgehip_tvrlds.so rebvo_amd/lib/libedgehip.so
for L in 0 19000 25000 39000 52000 79000; do
  echo -n "dyn LDS $L B/block: "
  EDGEHIP_TVR_LDS=$L timeout 300 python bench.py --steps 12 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "
import sys,json; l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); k=j['kernel_us_per_step']; print(j['value'], 'tvr', k['B.try_velrot'], 'per launch', j['roofline']['launch_us'])"
done
cp /tmp/keep.so rebvo_amd/lib/libedgehip.so
