"""Experiment: C contexts (streams) x nseq/C sequences vs one context — does kernel concurrency hide the serial
LM-step kernels and launch tails?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from rebvo_amd import edgehip, synth
W, H, POOL = 752, 480, 24
frames = [f for f, _, _ in synth.billboard_sequence(W, H, POOL, seed=11)]
pool = torch.from_numpy(np.stack(frames)).cuda()
def tri(k, n):
    p = 2 * (n - 1); k %= p
    return k if k < n else p - k
for spec in sys.argv[1:] or ["1x256", "2x128"]:
    C, B = (int(v) for v in spec.split("x"))
    TOT = C * B
    ehs = [edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=B, nslots=3) for _ in range(C)]
    offs = [(np.arange(B) + i * B) % (2 * (POOL - 1)) for i in range(C)]
    def step(k):
        for eh, o in zip(ehs, offs):
            idx = np.array([tri(k + x, POOL) for x in o], dtype=np.int32)
            eh.upload_rgb_indexed(eh.next_slot(), pool.data_ptr(), POOL, idx)
            eh.process_frame(0.05 * k)
    for k in range(12): step(k)
    for eh in ehs: eh.sync()
    t0 = time.perf_counter()
    K = 20
    for k in range(12, 12 + K): step(k)
    for eh in ehs: eh.sync()
    dt = time.perf_counter() - t0
    ok = sum(n.estimation_ok for eh in ehs for n in eh.read_nav())
    print(f"contexts={C} x nseq={B}: {TOT*K/dt:.0f} frames/s  ({dt/K*1e3:.3f} ms/step)  ok={ok}/{TOT}", flush=True)
    for eh in ehs: eh.close()
