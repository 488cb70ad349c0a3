"""Find what differs first between the HIP path and the reference on the pool-6 replay (diverges at frame 8)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rebvo_amd import edgehip, synth
from oracle import oracle
w, h, npool, nf = 752, 480, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 9
frames = [f for f, _, _ in synth.billboard_sequence(w, h, npool, seed=11)]
P2 = 2 * (npool - 1)
tri = lambda k: (k % P2) if (k % P2) < npool else P2 - (k % P2)
eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
for k in range(nf):
    f = frames[tri(k)]
    eh.upload_rgb(eh.next_slot(), f[None]); eh.process_frame(0.05 * k)
    ng = eh.read_nav()[0]
    _, nr = orc.process_frame(f, 0.05 * k)
    kg, mg = eh.download_keylines(0, eh.cur_slot())
    kr = orc.keylines(orc.cur_slot())
    mr = orc.mask(orc.cur_slot())
    line = [k, "kn", len(kg), len(kr), "dV %.2e" % np.abs(np.array(ng.V[:]) - np.array(nr.V[:])).max()]
    if len(kg) == len(kr):
        for fld in kg.dtype.names:
            a, b = kg[fld], kr[fld]
            if a.dtype.kind == "f":
                bad = ~(np.isclose(a, b, rtol=1e-9, atol=1e-12) | (np.isnan(a) & np.isnan(b)))
            else:
                bad = a != b
            if bad.ndim > 1: bad = bad.any(axis=tuple(range(1, bad.ndim)))
            if bad.any():
                line += [fld, int(bad.sum())]
    line += ["mask_eq", bool(np.array_equal(mg, mr))]
    print(*line, flush=True)
    if k >= nf - 2:
        # previous (old) slot state too
        so_g = (eh.cur_slot() - 1) % 3
        kg0, _ = eh.download_keylines(0, so_g)
        # oracle ring has 8 slots
        kr0 = orc.keylines((orc.cur_slot() - 1) % 8)
        l2 = ["  old slot:", len(kg0), len(kr0)]
        if len(kg0) == len(kr0):
            for fld in kg0.dtype.names:
                a, b = kg0[fld], kr0[fld]
                bad = ~(np.isclose(a, b, rtol=1e-9, atol=1e-12) | (np.isnan(a) & np.isnan(b))) if a.dtype.kind == "f" else a != b
                if bad.ndim > 1: bad = bad.any(axis=tuple(range(1, bad.ndim)))
                if bad.any(): l2 += [fld, int(bad.sum())]
        print(*l2)
