#!/bin/bash
# A/B two builds of libedgehip.so on the SAME box (boxes differ by 10-15 %): tools/experiments/bin/libedgehip_{A,B}.so
# are copied over rebvo_amd/lib/libedgehip.so in turn, ABAB, and bench.py's per-kernel times are printed.
cd "$GRAFT_REPO_ROOT"
for v in A B A B; do
  cp tools/experiments/bin/libedgehip_$v.so rebvo_amd/lib/libedgehip.so
  python bench.py --steps 10 --warmup 4 --no-extras --cpu-frames 0 > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"]), d["ms_per_step"], {k:round(x) for k,x in d["kernel_us_per_step"].items()})
PY
done
