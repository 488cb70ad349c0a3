#!/bin/bash
# one sequence, frames resident: kernel trace of a steady frame (where do 0.68 ms go?)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/single_trace; mkdir -p $OUT
cat > /tmp/single.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from rebvo_amd import edgehip, synth
w, h = 752, 480
frames = [f for f, _, _ in synth.billboard_sequence(w, h, 12)]
pool = torch.from_numpy(np.stack(frames).reshape(-1)).cuda()
pool = torch.cat([pool, torch.zeros(16, dtype=torch.uint8, device="cuda")])
tri = lambda k, n: (k % (2 * (n - 1))) if (k % (2 * (n - 1))) < n else 2 * (n - 1) - (k % (2 * (n - 1)))
eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
def run(k0, n):
    for k in range(k0, k0 + n):
        eh.bind_rgb_indexed(eh.next_slot(), pool.data_ptr(), 12, np.full(1, tri(k, 12), np.int32))
        eh.process_frame(0.05 * k)
    eh.sync()
run(0, 24)
t0 = time.perf_counter(); run(24, 60); print("ms/frame", (time.perf_counter() - t0) / 60 * 1e3)
PY
( cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace -o single -- python /tmp/single.py > $OUT/run.log 2>&1 )
tail -2 $OUT/run.log
DB=$(ls $OUT/trace/*.db $OUT/trace/*/*.db 2>/dev/null | head -1)
python tools/rocpd_timeline.py $DB k_rgb_rowscan > $OUT/timeline.txt; cat $OUT/timeline.txt
find $OUT -name '*.db' -delete
