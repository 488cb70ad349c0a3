"""bench.py's plugin-surface legs alone (host_surface), with the wall time of every leg's process."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from rebvo_amd import edgehip, synth
W, H = 752, 480
p = edgehip.euroc_params(W, H)
intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
t = time.time()
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 24, seed=11, **intr)]
print("render", round(time.time() - t, 1), "s", flush=True)
t = time.time()
hs = bench.host_surface(p, frames, W, H)
print("host_surface", round(time.time() - t, 1), "s")
for k, v in hs.get("detail", {}).items():
    print(k, v["fps"], "wall", v["process_wall_s"], "run", round(v["seconds"], 3), v.get("producer_us_per_frame"))
print(hs.get("errors"))
