import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_stereo_gpu as T
from rebvo_amd import edgehip
from oracle import oracle
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 30
p, frames, pairs, pc = T.make_data(all_pairs=True, nf=nf)
W, H = T.W, T.H
orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
orc.enable_stereo(pc["ppx"], pc["ppy"], pc["zfx"], pc["zfy"], T.T_PAIR, T.R_PAIR, 100.0)
eh = edgehip.EdgeHip(edgehip.euroc_params(W, H, stereo_available=1), nseq=2, nslots=4)
eh.set_slot_camera(3, pc["ppx"], pc["ppy"], pc["zfx"], pc["zfy"])
eh.set_stereo_rig(3, T.T_PAIR, T.R_PAIR, 100.0)
eh.set_nav_log(nf)
for k in range(nf):
    eh.upload_rgb(eh.next_slot(), np.stack([frames[k]] * 2))
    eh.upload_rgb(3, np.stack([pairs[k]] * 2))
    eh.process_frame(0.05 * k)
log = eh.read_nav_log(0, nf)
for k in range(nf):
    _, nr = orc.process_frame_stereo(frames[k], pairs[k], 0.05 * k)
    ng = log[k][0]
    if k == 0: continue
    print(k, "kn", ng.kn, nr.kn, "klm", ng.klm_num, nr.klm_num, "dV %.2e dW %.2e dPos %.2e" % (
        np.abs(np.array(ng.V[:]) - np.array(nr.V[:])).max(), np.abs(np.array(ng.W[:]) - np.array(nr.W[:])).max(),
        np.abs(np.array(ng.Pos[:]) - np.array(nr.Pos[:])).max()), "seqs equal", log[k][0].V[:] == log[k][1].V[:])
