"""Experiment: is the single-sequence path bound by the host's launch rate or by the device?  Time the enqueue loop alone
(process_frame returns when the launches are queued) and the loop + final synchronisation."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import torch

from rebvo_amd import edgehip, synth

w, h, P = 752, 480, 12
frames = [f for f, _, _ in synth.billboard_sequence(w, h, P)]
pool = torch.from_numpy(np.stack(frames).reshape(-1)).cuda()
pool = torch.cat([pool, torch.zeros(16, dtype=torch.uint8, device="cuda")])
tri = lambda k, n: (k % (2 * (n - 1))) if (k % (2 * (n - 1))) < n else 2 * (n - 1) - (k % (2 * (n - 1)))
for n in (1, 8):
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=n, nslots=3)
    idx = [np.full(n, tri(k, P), np.int32) for k in range(400)]

    def run(k0, cnt):
        for k in range(k0, k0 + cnt):
            eh.bind_rgb_indexed(eh.next_slot(), pool.data_ptr(), P, idx[k])
            eh.process_frame(0.05 * k)
    run(0, 24)
    eh.sync()
    for cnt in (1, 4, 16, 64, 200):
        eh.sync()
        t0 = time.perf_counter()
        run(24, cnt)
        t1 = time.perf_counter()
        eh.sync()
        t2 = time.perf_counter()
        print(f"n={n} frames={cnt:3d}: enqueue {1e3 * (t1 - t0) / cnt:.4f} ms/frame, with sync {1e3 * (t2 - t0) / cnt:.4f} ms/frame", flush=True)
    eh.close()
