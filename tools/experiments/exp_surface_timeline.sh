#!/bin/bash
# device timeline of the plugin surface at 64 objects: the step's copy against the frames' kernels (rocprofv3 kernel + memory-copy trace)
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/${1:-surface_timeline}; mkdir -p $OUT; export TMPDIR=/tmp
python - <<'PY'
import os, sys
sys.path.insert(0, '.')
import numpy as np
from rebvo_amd import config, edgehip, synth
W, H = 752, 480
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 12, seed=11)]
config.write_global_config('/tmp/cfg', edgehip.euroc_params(W, H))
np.stack(frames).tofile('/tmp/frames.rgb24')
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -o t -- $GRAFT_REPO_ROOT/rebvo_amd/lib/surface_replay /tmp/cfg /tmp/frames.rgb24 12 ${NOBJ:-64} 30 1 0.05 --warmup 6 --threads 8 --group g > $OUT/run.log 2>&1 )
tail -1 $OUT/run.log
find $OUT/tr -name "*.csv" | head
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
kf = glob.glob(out + '/tr/**/*kernel_trace.csv', recursive=True)[0]
mf = glob.glob(out + '/tr/**/*memory_copy_trace.csv', recursive=True)[0]
K = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kf))]
M = list(csv.DictReader(open(mf)))
print(M[0].keys())
big = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', ''), int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in M]
big = sorted(b for b in big if b[3] > 300e3)
K.sort()
t0 = big[len(big) // 2][0]
lo, hi = t0, big[min(len(big) - 1, len(big) // 2 + 4)][1]
ev = [(s, e, f"COPY {d}") for s, e, d, b in big if lo <= s <= hi]
# frames: mark the first kernel of each process_frame by name of stage A's first kernel = most common kernel following a long chain; just print every kernel > 40 us and idle gaps > 40 us
cur = None
for s, e, n in K:
    if s < lo or s > hi: continue
    n = n.replace('edgehip::', '').replace('void ', '').split('<')[0].split('(')[0]
    ev.append((s, e, n))
ev.sort()
last_end = lo
for s, e, n in ev:
    tag = ''
    if not n.startswith('COPY'):
        if s - last_end > 30e3: tag = f"   <-- compute idle {(s - last_end) / 1e3:.0f} us"
        last_end = max(last_end, e)
        if e - s < 60e3 and not tag: continue
    print(f"{(s - lo) / 1e3:9.0f} {(e - lo) / 1e3:9.0f} {(e - s) / 1e3:8.0f}  {n}{tag}")
PY
rm -rf $OUT/tr
