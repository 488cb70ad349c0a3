"""bench.py's `objects_8_imu_fps` leg alone, with the group thread's own accounting (REBVO_GROUP_TIMING=1)."""
import json, os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from rebvo_amd import edgehip, synth, config
W, H = 752, 480
p = edgehip.euroc_params(W, H)
intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 24, seed=11, **intr)]
td = tempfile.mkdtemp()
np.stack(frames).tofile(td + "/f.rgb24")
n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 300
bench._write_surface_imu_csv(td + "/imu.csv", 24, k + n + 2, 1.0, 0.05)
config.write_global_config(td + "/cfg", p, imu=dict(mode=2, file=td + "/imu.csv", time_scale=1.0, InitBiasFrameNum=3))
for extra in ([], ["--callback"]):
    r = subprocess.run([os.path.join(bench.ROOT, "rebvo_amd/lib/surface_replay"), td + "/cfg", td + "/f.rgb24", "24", str(n), str(k), "1", "0.05", "--warmup", "40",
                        "--threads", str(min(16, n)), "--group", "g", "--stagger"] + extra, capture_output=True, text=True, timeout=200,
                       env=dict(os.environ, REBVO_GROUP_TIMING="1"))
    print(extra, r.stdout.strip().splitlines()[-1][:300])
    print("   ", [l for l in r.stdout.splitlines() if "us per step" in l][-1][:420])
