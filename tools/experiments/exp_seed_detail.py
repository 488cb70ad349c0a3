#!/usr/bin/env python3
"""One scene of exp_seed_sweep.py frame by frame: |dV|, |dW| free-running and teacher-forced, and the reference's knife-edge frames
(oracle/teacher.py).  usage: exp_seed_detail.py W H SEED NPOOL NFRAMES PHASE"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rebvo_amd import edgehip, synth
from oracle import oracle, teacher

w, h, seed, npool, nf = (int(x) for x in sys.argv[1:6])
phases = [int(x) for x in sys.argv[6:]] or [0]


def tri(k, n):
    p = 2 * (n - 1); k %= p
    return k if k < n else p - k


frames = [f for f, _, _ in synth.billboard_sequence(w, h, npool, seed=seed)]
for ph in phases:
    for forced in (False, True):
        orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
        r = teacher.teacher_forced_replay(eh, orc, lambda k: frames[tri(k + ph, npool)], nf, forced=forced)
        eh.close(); orc.close()
        print(f"phase {ph} {'teacher-forced' if forced else 'free-running '}: dV", " ".join(f"{x:.1e}" for x in r["dV"]))
        print(f"                              dW", " ".join(f"{x:.1e}" for x in r["dW"]))
        print("   outside tolerance:", r["outside_tolerance"], " knife-edge frames:", [(f["frame"], len(f["keylines"])) for f in r["knife_edge_frames"]])
