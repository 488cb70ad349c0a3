#!/bin/bash
# plugin surface, 64 objects in one group: frames/s against the number of producer threads (is the application's copyFrom the bound?)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, subprocess, json, tempfile
sys.path.insert(0, '.')
import numpy as np
import bench
from rebvo_amd import config, edgehip
from rebvo_amd import synth
W, H = 752, 480
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 12, seed=11)]
params = edgehip.euroc_params(W, H)
exe = 'rebvo_amd/lib/surface_replay'
with tempfile.TemporaryDirectory() as td:
    cfg, raw = td + '/cfg', td + '/frames.rgb24'
    config.write_global_config(cfg, params)
    np.stack(frames).tofile(raw)
    import ast
    for n, k, wm in ast.literal_eval(os.environ.get('CASES', '((64, 36, 6), (128, 30, 6))')):
        for t in ast.literal_eval(os.environ.get('THREADS', '(4, 8, 16, 32, 64)')):
            for rep in range(int(os.environ.get('REPS', '2'))):
                r = subprocess.run([exe, cfg, raw, str(len(frames)), str(n), str(k), '1', str(bench.FRAME_DT), '--warmup', str(wm), '--threads', str(t), '--group', 'g'] + os.environ.get('EXTRA', '').split(),
                                   capture_output=True, text=True, timeout=120)
                try:
                    js = json.loads(r.stdout.strip().splitlines()[-1])
                    print(n, 'objects', t, 'threads', js['fps'], js['ms_per_step'], flush=True)
                    for l in r.stdout.splitlines():
                        if 'us per step on the group thread' in l or l.startswith('  step '): print('   ', l, flush=True)
                except Exception as e:
                    print(n, t, 'failed', r.returncode, r.stdout[-200:], flush=True)
PY
nproc; grep -m1 "model name" /proc/cpuinfo
