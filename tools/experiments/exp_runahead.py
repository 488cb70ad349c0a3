import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rebvo_amd import edgehip, synth
from oracle import oracle
w, h, npool, nf, B = 752, 480, 6, 14, 4
frames = [f for f, _, _ in synth.billboard_sequence(w, h, npool, seed=11)]
host = np.stack(frames)
pool = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
pool[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
torch.cuda.synchronize()
P2 = 2 * (npool - 1)
tri = lambda k: (k % P2) if (k % P2) < npool else P2 - (k % P2)
UNEVEN = int(os.environ.get("UNEVEN", "1"))
t = lambda k: 0.05 * k + (0.003 * (k % 3) if UNEVEN else 0.0)
logs = []
for sync in (0, 1):
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=B, nslots=3)
    eh.set_nav_log(nf)
    for k in range(nf):
        idx = np.array([tri(k + s) for s in range(B)], np.int32)
        eh.bind_rgb_indexed(eh.next_slot(), pool.data_ptr(), npool, idx)
        eh.process_frame(t(k))
        if sync: eh.sync()
    logs.append(eh.read_nav_log(0, nf))
    eh.close()
for s in range(B):
    for k in range(nf):
        a, b = logs[0][k][s], logs[1][k][s]
        if a.V[:] != b.V[:] or a.kn != b.kn:
            print("nosync != sync at seq", s, "frame", k); break
for s in (0,):
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
    for k in range(nf):
        _, nr = orc.process_frame(frames[tri(k + s)], t(k))
        ng = logs[1][k][s]
        print(k, "pool", tri(k+s), "kn", ng.kn, nr.kn, "dV %.2e" % np.abs(np.array(ng.V[:]) - np.array(nr.V[:])).max(), "ok", ng.estimation_ok, nr.estimation_ok, "score %.6e %.6e" % (ng.score, nr.score), "evals", ng.minimizer_evals)
