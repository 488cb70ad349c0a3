"""Eight ImuMode=2 sequences through the C-ABI with the batch group's call pattern: where does a step's 1.1 ms go?"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rebvo_amd import edgehip, synth
W, H, B, K = 752, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 200
p = edgehip.euroc_params(W, H)
intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 8, seed=11, **intr)]
grey = [np.ascontiguousarray(f[:, :, 0]) for f in frames]
def tri(k, n):
    q = 2 * (n - 1); k %= q
    return k if k < n else q - k
def imu_rec():
    r = edgehip.ImuIntegrated(); r.n = 10; r.dt = 0.005
    r.Rot[:] = np.eye(3).reshape(-1); r.giro[:] = [0.004, -0.002, 0.003]; r.acel[:] = [0, -9.8, 0]; r.cacel[:] = [0, -9.8, 0]
    return r
for mode in ("no reads", "nav log of k-2", "nav + imu log of k-2", "nav + imu log of k-1", "nav + imu log of k"):
    eh = edgehip.EdgeHip(p, nseq=B, nslots=3)
    eh.imu_enable(edgehip.euroc_imu_params(init_bias_frame_num=3))
    eh.set_nav_log(8)
    recs = [imu_rec() for _ in range(B)]
    t0 = None
    for k in range(K + 20):
        if k == 20:
            eh.sync(); t0 = time.perf_counter()
        eh.upload_grey8(eh.next_slot(), np.stack([grey[tri(k + s, 8)] for s in range(B)]))
        eh.set_imu(recs)
        eh.process_frame(np.full(B, 1.0 + 0.05 * k))
        lag = {"no reads": None, "nav log of k-2": 2, "nav + imu log of k-2": 2, "nav + imu log of k-1": 1, "nav + imu log of k": 0}[mode]
        if lag is not None and k - lag >= 0:
            eh.read_nav_log_array(k - lag, 1)
            if "imu" in mode:
                eh.read_nav_imu_log(k - lag, 1)
    eh.sync()
    print(f"{mode:24s} {(time.perf_counter() - t0) / K * 1e3:.3f} ms per step", flush=True)
    eh.close()
