"""Heterogeneous batch of bench.py (six scenes, phases, scene cuts) at a small batch size: every checked sequence frame by
frame against the CPU reference, to see where (if anywhere) the HIP path leaves it."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from oracle import oracle
from rebvo_amd import edgehip, synth
from bench import tri

w, h, S, PF, n, Wm, K = 752, 480, 6, 12, int(sys.argv[1]) if len(sys.argv) > 1 else 48, 24, 40
p = edgehip.euroc_params(w, h)
intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
scenes = [[f for f, _, _ in synth.billboard_sequence(w, h, PF, seed=101 + 7 * s, traj_seed=29 + s, **intr)] for s in range(S)]
hframes = [f for sc in scenes for f in sc]
host = np.stack(hframes)
pool = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
pool[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
scene_of = np.arange(n) % S
phase = (np.arange(n) // S) % (2 * (PF - 1))
cut = (np.arange(n) % 16) == 5
kcut = Wm + K // 2


def hidx(k):
    sc = np.where(cut & (k >= kcut), (scene_of + 1) % S, scene_of)
    return sc * PF + np.array([tri(k + q, PF) for q in phase])


eh = edgehip.EdgeHip(p, nseq=n, nslots=3)
eh.set_nav_log(Wm + K)
for k in range(Wm + K):
    eh.bind_rgb_indexed(eh.next_slot(), pool.data_ptr(), S * PF, hidx(k).astype(np.int32))
    eh.process_frame(0.05 * k)
log = eh.read_nav_log_array(0, Wm + K)
for s in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,5,1,2,3,4")]:
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
    worst = 0.0
    for k in range(Wm + K):
        _, nr = orc.process_frame(hframes[int(hidx(k)[s])], 0.05 * k)
        g = log[k, s]
        dv = float(np.max(np.abs(g["V"] - np.array(nr.V[:])))) if np.all(np.isfinite(nr.V[:])) else 0.0
        dw = float(np.max(np.abs(g["W"] - np.array(nr.W[:])))) if np.all(np.isfinite(nr.W[:])) else 0.0
        step = np.linalg.norm(nr.V[:]) + np.linalg.norm(nr.W[:])
        flag = ""
        if g["kn"] != nr.kn or g["estimation_ok"] != nr.estimation_ok or dv > 1e-6 * step + 1e-9 or dw > 1e-6 * step + 1e-9:
            flag = "  <-- differs"
        if flag or k in (0, kcut - 1, kcut, kcut + 1, Wm + K - 1):
            print(f"seq {s} frame {k}: kn {g['kn']}/{nr.kn} ok {g['estimation_ok']}/{nr.estimation_ok} klm {g['klm_num']}/{nr.klm_num} "
                  f"|V| {np.linalg.norm(nr.V[:]):.3e} |W| {np.linalg.norm(nr.W[:]):.3e} dV {dv:.2e} dW {dw:.2e} score {g['score']:.6g}/{nr.score:.6g}{flag}")
        worst = max(worst, dv, dw)
    print(f"seq {s}: worst |dV|,|dW| = {worst:.3e}")
    orc.close()
