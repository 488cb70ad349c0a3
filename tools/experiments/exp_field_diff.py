import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
from rebvo_amd import edgehip, synth
from oracle import oracle
from helpers import inject_pair
w, h, npool = 752, 480, 6
NW = 8
frames = [f for f, _, _ in synth.billboard_sequence(w, h, npool, seed=11)]
P2 = 2 * (npool - 1)
tri = lambda k: (k % P2) if (k % P2) < npool else P2 - (k % P2)
orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
for k in range(NW):
    _, nav = orc.process_frame(frames[tri(k)], 0.05 * k)
so, sn = (NW - 1) % 8, NW % 8
orc.stage_a(sn, frames[tri(NW)], nav.tresh, nav.kn)
eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=2)
inject_pair(eh, orc, so, sn)
orc.build_field(sn, 40, orc.retuned(sn)); eh.build_field(1, 40, -1.0)
fr, fg = orc.field(sn), eh.download_field(0)
bad = (fr[..., 1] != fg[..., 1]) | ((fr[..., 1] >= 0) & (fr[..., 0] != fg[..., 0]))
ys, xs = np.nonzero(bad)
print("differing pixels", len(ys), "retuned", orc.retuned(sn))
kl = orc.keylines(sn)
for y, x in list(zip(ys, xs))[:20]:
    r, g = fr[y, x], fg[y, x]
    print((x, y), "ref dist,ikl", r, "gpu", g)
    for ik in {int(r[1]), int(g[1])}:
        if ik >= 0:
            k = kl[ik]
            print("    kl", ik, "c_p", k["c_p"], "u_m", k["u_m"], "n_m", k["n_m"])
