import sys, time
sys.path.insert(0, '.')
import numpy as np
from rebvo_amd import edgehip, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = int(sys.argv[2]) if len(sys.argv) > 2 else 752
H = int(sys.argv[3]) if len(sys.argv) > 3 else 480
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 3)]
eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=B, nslots=2)
for k in range(2):
    eh.upload_rgb(k, np.stack([frames[k]] * B))
for it in range(3):
    eh.stage_a(it % 2)
eh.sync()
N = 20
t0 = time.time()
for it in range(N):
    eh.stage_a(it % 2)
eh.sync()
dt = (time.time() - t0) / N
print(f"B={B}: stage A {dt*1e6:.1f} us/step -> {B/dt:.0f} frames/s, kn={eh.get_kn(1)[:4]}")
eh.profile_enable(True)
for it in range(N):
    eh.stage_a(it % 2)
pr = eh.profile_read()
for k, (ms, calls) in pr.items():
    if calls:
        print(f"  {k:20s} {ms/calls*1e3:9.1f} us/call  x{calls//N}/frame  {ms/N*1e3:9.1f} us/frame")
