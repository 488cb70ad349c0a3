#!/usr/bin/env python3
"""Whole path against the reference over many random scenes (seeds the test suite does not use): per seed a batch of three sequences at
different phases of a frame pool, 20 frames, every frame's record compared with the CPU reference fed the same frames (the bounds of
tests/test_soak_gpu.py).  Prints one line per seed and the departures, if any.  A measurement, not a test."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rebvo_amd import edgehip, synth
from oracle import oracle


def tri(k, n):
    p = 2 * (n - 1); k %= p
    return k if k < n else p - k


def run(w, h, seed, npool, nf, phases, dt=0.05, tum=False, over=None, **intr):
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, npool, seed=seed, **intr)]
    over = over or {}
    gp, op = ((edgehip.tum_params(w, h, use_undistort=1), oracle.tum_params(w, h, use_undistort=1)) if tum
              else (edgehip.euroc_params(w, h, **over), oracle.euroc_params(w, h, **over)))
    eh = edgehip.EdgeHip(gp, nseq=len(phases), nslots=3)
    eh.set_nav_log(nf)
    for k in range(nf):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[tri(k + p, npool)] for p in phases]))
        eh.process_frame(dt * k)
    log = eh.read_nav_log(0, nf)
    eh.close()
    bad = []
    worst = 0.0
    for s, p in enumerate(phases):
        orc = oracle.Oracle("ref", op)
        for k in range(nf):
            _, nr = orc.process_frame(frames[tri(k + p, npool)], dt * k)
            ng = log[k][s]
            if (ng.kn, ng.estimation_ok, ng.klm_num) != (nr.kn, nr.estimation_ok, nr.klm_num):
                bad.append((p, k, "counts", (ng.kn, ng.estimation_ok, ng.klm_num), (nr.kn, nr.estimation_ok, nr.klm_num)))
                break
            if k == 0:
                continue
            d = max(np.abs(np.array(ng.V[:]) - np.array(nr.V[:])).max(), np.abs(np.array(ng.W[:]) - np.array(nr.W[:])).max())
            worst = max(worst, d)
            if d > 1e-9:
                bad.append((p, k, "dVW", d))
                break
    return bad, worst


t0 = time.time()
total_bad = 0
for (w, h, seeds, npool, nf) in (() if len(sys.argv) > 1 and sys.argv[1] == "more" else ((376, 240, range(100, 124), 10, 20), (752, 480, range(200, 208), 8, 14), (640, 480, range(300, 306), 8, 14))):
    for seed in seeds:
        bad, worst = run(w, h, seed, npool, nf, (0, 3, 5))
        total_bad += len(bad)
        print(f"{w}x{h} seed {seed}: worst |dV|,|dW| {worst:.2e}  departures {bad}", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "more":
    # GlobalConfig_desk.txt values with the undistortion map (TUM intrinsics), and random parameter variants of the EuRoC values
    for seed in range(400, 410):
        bad, worst = run(640, 480, seed, 8, 24, (0, 4), dt=0.02, tum=True, fx=525.0, fy=525.0, cx=320.0, cy=240.0)
        total_bad += len(bad)
        print(f"TUM + undistort 640x480 seed {seed}: worst |dV|,|dW| {worst:.2e}  departures {bad}", flush=True)
    rng = np.random.default_rng(7)
    for seed in range(500, 516):
        over = dict(tracker_iter_num=int(rng.integers(1, 13)), search_range=int(rng.integers(10, 60)), qcut_quantile=float(rng.uniform(0.5, 0.95)),
                    match_num_thresh=int(rng.integers(0, 5)), regularize_thresh=float(rng.uniform(0.2, 0.8)), reweight_distance=float(rng.uniform(1.0, 3.0)),
                    tracker_init_type=int(rng.integers(0, 3)))   # (not DetectorPlaneFitSize: the reference keeps the fit's pseudo-inverse in a static sized by the first call of the process)
        try:
            bad, worst = run(376, 240, seed, 10, 16, (0, 3), over=over)
        except Exception as e:
            bad, worst = [("error", str(e)[:120])], float("nan")
        total_bad += len(bad)
        print(f"variant seed {seed} {over}: worst {worst:.2e}  departures {bad}", flush=True)
print(f"{total_bad} departures, {time.time() - t0:.0f} s")
