"""Frames the tracker cannot make sense of (white noise, black, a scene cut every frame, constant image): every frame must
still finish in bounded time.  Prints the time of each process_frame + sync and the slowest kernel group."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from rebvo_amd import edgehip, synth
w, h, n = 752, 480, 8
rs = np.random.RandomState(3)
scene = [f for f, _, _ in synth.billboard_sequence(w, h, 4)]
def frame(kind, k):
    if kind == "noise": return rs.randint(0, 256, (n, h, w, 3), dtype=np.uint8)
    if kind == "black": return np.zeros((n, h, w, 3), np.uint8)
    if kind == "const": return np.full((n, h, w, 3), 200, np.uint8)
    if kind == "cuts":  return np.stack([synth.billboard_sequence(w, h, 1, seed=1000 + 17 * k + s)[0][0] if False else np.roll(scene[(k + s) % 4], 97 * k * (s + 1), axis=1) for s in range(n)])
    if kind == "mixed": return np.stack([scene[k % 4] if (k + s) % 3 else rs.randint(0, 256, (h, w, 3), dtype=np.uint8) for s in range(n)])
for kind in ("noise", "black", "const", "cuts", "mixed"):
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=n, nslots=3)
    eh.profile_enable(True)
    worst = 0.0
    for k in range(10):
        eh.upload_rgb(eh.next_slot(), np.ascontiguousarray(frame(kind, k)))
        t0 = time.time()
        eh.process_frame(0.05 * k)
        eh.sync()
        dt = time.time() - t0
        worst = max(worst, dt)
    pr = eh.profile_read()
    top = sorted(((v[0], k2) for k2, v in pr.items() if v[1]), reverse=True)[:3]
    nav = eh.read_nav()
    print(f"{kind:6s} worst frame {worst * 1e3:8.2f} ms   kn {[int(x.kn) for x in nav][:4]} ok {[int(x.estimation_ok) for x in nav][:4]}  top groups (ms over 10 frames): {[(k2, round(ms, 2)) for ms, k2 in top]}", flush=True)
    eh.close()
