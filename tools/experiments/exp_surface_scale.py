"""Frames/s through the rebvo::REBVO surface for large groups: surface_replay with N objects in one group."""
import json, os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rebvo_amd import edgehip, synth, config
W, H = 752, 480
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exe = os.path.join(ROOT, "rebvo_amd", "lib", "surface_replay")
td = tempfile.mkdtemp()
p = edgehip.euroc_params(W, H)
intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 24, seed=11, **intr)]
np.stack(frames).tofile(td + "/f.rgb24")
config.write_global_config(td + "/cfg", p)
for n, k, wm, th in [tuple(int(x) for x in a.split(":")) for a in (sys.argv[1:] or ["64:240:40:16", "256:120:30:16", "1024:60:20:16"])]:
    r = subprocess.run([exe, td + "/cfg", td + "/f.rgb24", "24", str(n), str(k), "1", "0.05", "--warmup", str(wm), "--threads", str(th), "--group", f"g{n}"],
                       capture_output=True, text=True, timeout=150, env=dict(os.environ, REBVO_GROUP_TIMING="1"))
    try:
        js = json.loads(r.stdout.strip().splitlines()[-1])
        print(n, "objects", th, "threads:", js["fps"], "frames/s", js["ms_per_step"], "ms/step", flush=True)
        print("   ", [l for l in r.stdout.splitlines() if "us per step" in l][-1][:400], flush=True)
    except Exception as e:
        print(n, "failed", r.returncode, (r.stdout + r.stderr)[-400:], flush=True)
