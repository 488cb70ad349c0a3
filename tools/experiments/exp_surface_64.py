#!/usr/bin/env python3
"""The plugin surface's 64-object leg (8-bit mono steps) under the one-kernel stage A's dispatch rule: default, EDGEHIP_FUSED_MIN_BATCH = 32, 192."""
import json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rebvo_amd import config, edgehip, synth
W, H = 752, 480
frames = list(synth.rects_sequence(W, H, 24))
exe = "rebvo_amd/lib/surface_replay"
with tempfile.TemporaryDirectory() as td:
    cfg, raw = td + "/cfg", td + "/frames.rgb24"
    config.write_global_config(cfg, edgehip.euroc_params(W, H))
    np.stack(frames).tofile(raw)
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        for mb in (None, "192"):
            env = dict(os.environ)
            if mb: env["EDGEHIP_FUSED_MIN_BATCH"] = mb
            r = subprocess.run([exe, cfg, raw, "24", "64", "240", "1", "0.05", "--warmup", "40", "--threads", "16", "--group", f"g{rep}{mb}"], env=env, capture_output=True, text=True, timeout=120)
            js = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stdout + r.stderr)[-300:]}
            print("min_batch", mb or "default", js.get("fps"), js.get("error", ""), flush=True)
