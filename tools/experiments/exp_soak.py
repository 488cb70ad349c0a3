"""Soak: long replays of several scenes / parameter sets, HIP path (batch of phases, no host sync) vs the CPU reference."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rebvo_amd import edgehip, synth
from oracle import oracle

def tri(k, n):
    p = 2 * (n - 1); k = k % p
    return k if k < n else p - k

def run(name, frames, gp, op, nf, phases, dt=0.05):
    npool = len(frames)
    B = len(phases)
    eh = edgehip.EdgeHip(gp, nseq=B, nslots=3)
    eh.set_nav_log(nf)
    for k in range(nf):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[tri(k + p, npool)] for p in phases]))
        eh.process_frame(dt * k)
    log = eh.read_nav_log(0, nf)
    eh.close()
    for s, p in enumerate(phases):
        orc = oracle.Oracle("ref", op)
        worst, first = 0.0, None
        for k in range(nf):
            _, nr = orc.process_frame(frames[tri(k + p, npool)], dt * k)
            ng = log[k][s]
            d = max(np.abs(np.array(ng.V[:]) - np.array(nr.V[:])).max(), np.abs(np.array(ng.W[:]) - np.array(nr.W[:])).max())
            if ng.kn != nr.kn or ng.estimation_ok != nr.estimation_ok: d = max(d, 1.0)
            if d > 1e-9 and first is None: first = (k, d, ng.kn, nr.kn, ng.estimation_ok, nr.estimation_ok)
            worst = max(worst, d)
        print(name, "phase", p, "frames", nf, "worst %.2e" % worst, "first divergence", first, flush=True)

which = sys.argv[1:] or ["bb6", "rects", "tum", "small", "bb24"]
if "bb6" in which:
    fr = [f for f, _, _ in synth.billboard_sequence(752, 480, 6, seed=11)]
    run("euroc billboard pool6", fr, edgehip.euroc_params(752, 480), oracle.euroc_params(752, 480), 40, [0, 1, 3])
if "rects" in which:
    fr = list(synth.rects_sequence(752, 480, 16))
    run("euroc rects pool16", fr, edgehip.euroc_params(752, 480), oracle.euroc_params(752, 480), 40, [0, 5])
if "tum" in which:
    fr = [f for f, _, _ in synth.billboard_sequence(640, 480, 10, fx=525.0, fy=525.0, cx=320.0, cy=240.0, seed=3)]
    run("tum undistort pool10", fr, edgehip.tum_params(640, 480, use_undistort=1), oracle.tum_params(640, 480, use_undistort=1), 40, [0, 4], dt=0.02)
if "small" in which:
    fr = [f for f, _, _ in synth.billboard_sequence(376, 240, 12, seed=5)]
    run("euroc 376x240 pool12", fr, edgehip.euroc_params(376, 240), oracle.euroc_params(376, 240), 80, [0, 2, 7])
if "bb24" in which:
    fr = [f for f, _, _ in synth.billboard_sequence(752, 480, 24, seed=12)]
    run("euroc billboard pool24 seed12", fr, edgehip.euroc_params(752, 480), oracle.euroc_params(752, 480), 60, [0, 11])
