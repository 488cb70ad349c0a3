"""Experiment: latency of the full path at 1 / 8 / 64 sequences per launch (the numbers of bench.py's batch_sweep), with a
digest of the nav records so that two libraries can be compared for identical results.
  EDGEHIP_LIB=<path> python tools/experiments/exp_small_batch.py [n ...]"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import torch

import bench
from rebvo_amd import edgehip, synth

w, h, P = 752, 480, 24
params = edgehip.euroc_params(w, h)
intr = dict(fx=float(params.zfx), fy=float(params.zfy), cx=float(params.ppx), cy=float(params.ppy))
frames = [f for f, _, _ in synth.billboard_sequence(w, h, P, seed=11, **intr)]
host = np.stack(frames)
pool = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
pool[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
torch.cuda.synchronize()
for n in [int(a) for a in sys.argv[1:]] or [1, 8, 64]:
    o = np.arange(n, dtype=np.int64) % (2 * (P - 1))
    best = None
    for rep in range(3):
        rp = bench.Replay(edgehip, params, n, pool, P, lambda k, o=o: [bench.tri(k + x, P) for x in o], 0)
        k2 = 200 if n == 1 else 60
        dt, _ = bench.timed_replay(rp, k2, 12)
        nav = rp.ehs[0].read_nav()
        dig = hashlib.sha1(b"".join(bytes(x) for x in nav)).hexdigest()[:12]
        rp.close()
        best = dt if best is None else min(best, dt)
    print(f"n={n:4d}  ms_per_step {best / k2 * 1e3:.4f}  frames/s {n * k2 / best:.1f}  nav digest {dig}", flush=True)
