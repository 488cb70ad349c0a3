#!/usr/bin/env python3
"""The one-kernel stage A's batch threshold at image sizes that take its generic (run-time width) instantiation: frames/s of the whole path at B sequences,
EDGEHIP_FUSED_MIN_BATCH = 192 (multi-kernel stage A) against 32.  usage: exp_threshold_other_sizes.py W H B"""
import os, subprocess, sys, time
if len(sys.argv) > 4:      # child
    sys.path.insert(0, os.getcwd())
    import numpy as np
    from rebvo_amd import edgehip, synth
    W, H, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    pool = [f for f, _, _ in synth.billboard_sequence(W, H, 8, seed=4)]
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=B, nslots=3)
    frames = [np.stack([pool[(k + s) % len(pool)] for s in range(B)]) for k in range(8)]
    def step(k):
        eh.upload_rgb(eh.next_slot(), frames[k % 8]); eh.process_frame(0.05 * k)
    for k in range(30): step(k)
    eh.sync(); t0 = time.time()
    N = 100
    for k in range(30, 30 + N): step(k)
    eh.sync(); dt = (time.time() - t0) / N
    print(f"{B / dt:9.0f} frames/s  {dt * 1e3:.3f} ms per step  kn {eh.get_kn(eh.cur_slot())[0]}")
    sys.exit(0)
W, H, B = sys.argv[1:4]
for mb in ("192", "32", "default"):
    env = dict(os.environ, EDGEHIP_FUSED_MIN_BATCH=mb)
    if mb == "default": env.pop("EDGEHIP_FUSED_MIN_BATCH")
    r = subprocess.run([sys.executable, __file__, W, H, B, "child"], env=env, capture_output=True, text=True)
    print(f"{W}x{H} B={B} min_batch {mb}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
