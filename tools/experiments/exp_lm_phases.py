import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from rebvo_amd import edgehip, synth
w, h = 752, 480
frames = [f for f, _, _ in synth.billboard_sequence(w, h, 6)]
eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
for k in range(5):
    eh.upload_rgb(eh.next_slot(), frames[k][None])
    eh.process_frame(0.05 * k)
    eh.sync()
    print("---- frame", k, flush=True)
eh.close()
