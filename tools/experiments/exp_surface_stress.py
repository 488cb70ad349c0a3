#!/usr/bin/env python3
"""Random shapes of a batch group through surface_replay: group size, run length, producer threads, with / without callbacks, a member
that leaves, a dropped frame, a tinted frame, MonoUpload on / off — looking for hangs (time-out 60 s) and non-zero exits.  Not a test:
a robustness sweep (results: every run's last JSON line)."""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rebvo_amd import config, edgehip, synth

W, H = 376, 240
exe = "rebvo_amd/lib/surface_replay"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 8, seed=41)]
NMIN = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # group sizes: 1..16 by default; 30..80 takes the groups through the one-kernel stage A (from 32 members on)
NMAX = int(sys.argv[4]) if len(sys.argv) > 4 else 16
STEREO_P = float(sys.argv[5]) if len(sys.argv) > 5 else 0.3
bad = 0
with tempfile.TemporaryDirectory() as td:
    np.stack(frames).tofile(td + "/frames.rgb24")
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        n = int(rng.integers(NMIN, NMAX + 1)); k = int(rng.integers(4, 40)); t = int(rng.integers(1, min(n, 8) + 1))
        mono = int(rng.integers(0, 2))
        cfg = f"{td}/cfg{it}"
        gpu = dict(mono=mono)
        stereo = rng.random() < STEREO_P                                # (round 6) StereoAvaiable members: a pair frame per main frame (the pair images = the main images)
        if not stereo and rng.random() < 0.25: gpu["tracker_precision"] = 32          # (round 6) the float tracker behind the surface (refused with a stereo rig)
        pp = edgehip.euroc_params(W, H)
        config.write_global_config(cfg, pp, gpu=gpu, **(dict(dataset=("unused/", "unused.csv", 1.0), stereo=dict(dir="unused/", file="unused.csv", ppx=pp.ppx + 3,
                                                                                                              ppy=pp.ppy + 2, zfx=pp.zfx, zfy=pp.zfy)) if stereo else {}))
        args = [exe, cfg, td + "/frames.rgb24", "8", str(n), str(k), "1", "0.05", "--threads", str(t), "--group", f"s{it}"]
        if stereo: args += ["--stereo", td + "/frames.rgb24"]
        if rng.random() < 0.5: args.append("--callback")
        if n > 1 and rng.random() < 0.5: args += ["--leave", f"{int(rng.integers(0, n))}:{int(rng.integers(1, k))}"]
        if rng.random() < 0.4: args += ["--dup", f"{int(rng.integers(0, n))}:{int(rng.integers(1, k))}"]
        if rng.random() < 0.4: args += ["--tint", f"{int(rng.integers(0, n))}:{int(rng.integers(0, k))}"]
        if rng.random() < 0.2: args += ["--snapshot-at", str(int(rng.integers(0, k)))]
        try:
            r = subprocess.run(args, capture_output=True, text=True, timeout=60, cwd=os.getcwd())
            last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
            ok = r.returncode == 0 and last.startswith("{")
            print(("ok  " if ok else "BAD ") + " ".join(args[4:]), "| rc", r.returncode, "|", last[:110] if ok else (r.stdout + r.stderr)[-300:], flush=True)
            bad += 0 if ok else 1
        except subprocess.TimeoutExpired:
            print("HANG " + " ".join(args[4:]), flush=True)
            bad += 1
print(f"{bad} bad runs")
