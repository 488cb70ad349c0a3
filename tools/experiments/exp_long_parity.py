"""Per-frame HIP-vs-reference deltas over a long replay of bench.py's sequence 0 (pool of 24 frames, triangle order)."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rebvo_amd import edgehip, synth
from oracle import oracle
W, H = 752, 480
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pool = 24
def tri(k, n):
    p = 2 * (n - 1); k = k % p
    return k if k < n else p - k
frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=11)]
eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=1, nslots=3, device=0)
orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
for k in range(NF):
    f = frames[tri(k, pool)]
    eh.upload_rgb(eh.next_slot(), f[None])
    eh.process_frame(0.05 * k)
    ng = eh.read_nav()[0]
    _, nr = orc.process_frame(f, 0.05 * k)
    dV = np.abs(np.array(ng.V[:]) - np.array(nr.V[:])).max(); dW = np.abs(np.array(ng.W[:]) - np.array(nr.W[:])).max()
    print(k, "kn", ng.kn, nr.kn, "klm", ng.klm_num, nr.klm_num, "fwd", ng.klm_fwd, nr.klm_fwd, "ok", ng.estimation_ok, nr.estimation_ok,
          "dV %.2e dW %.2e" % (dV, dW), "Kp %.3e %.3e" % (ng.Kp, nr.Kp), "tresh %.6f %.6f" % (ng.tresh, nr.tresh),
          "dPos %.2e" % np.abs(np.array(ng.Pos[:]) - np.array(nr.Pos[:])).max(), flush=True)
