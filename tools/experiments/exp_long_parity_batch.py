"""bench.py's setup in small: B sequences at different pool phases, frames bound in place, nav log read back at the end;
sequence s against the CPU reference run on the same frame order."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rebvo_amd import edgehip, synth
from oracle import oracle
W, H = 752, 480
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
FIRST = int(sys.argv[3]) if len(sys.argv) > 3 else 0
P = 24
def tri(k, n):
    p = 2 * (n - 1); k = k % p
    return k if k < n else p - k
frames = [f for f, _, _ in synth.billboard_sequence(W, H, P, seed=11)]
host_pool = np.stack(frames)
pool = torch.empty(host_pool.size + 16, dtype=torch.uint8, device="cuda")
pool[:host_pool.size] = torch.from_numpy(host_pool.reshape(-1)).cuda()
torch.cuda.synchronize()
eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=B, nslots=3, device=0)
L = NF - FIRST
eh.set_nav_log(L)
off = np.arange(B) % (2 * (P - 1))
for k in range(NF):
    idx = np.array([tri(k + o, P) for o in off], dtype=np.int32)
    if os.environ.get("EXP_COPY"):
        eh.upload_rgb_indexed(eh.next_slot(), pool.data_ptr(), P, idx)
    else:
        eh.bind_rgb_indexed(eh.next_slot(), pool.data_ptr(), P, idx)
    eh.process_frame(0.05 * k)
    if os.environ.get("EXP_SYNC"):
        eh.sync()
log = eh.read_nav_log(FIRST, L)
for s in (0, 1, B - 1):
    orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
    worst = 0.0
    for k in range(NF):
        _, nr = orc.process_frame(frames[tri(k + int(off[s]), P)], 0.05 * k)
        if k < FIRST:
            continue
        ng = log[k - FIRST][s]
        d = max(np.abs(np.array(ng.V[:]) - np.array(nr.V[:])).max(), np.abs(np.array(ng.W[:]) - np.array(nr.W[:])).max(),
                np.abs(np.array(ng.Pos[:]) - np.array(nr.Pos[:])).max())
        if d > 1e-9 and worst <= 1e-9:
            print("seq", s, "first divergence at frame", k, "d", d, "kn", ng.kn, nr.kn, "frame field", ng.frame, nr.frame)
        worst = max(worst, d)
    print("seq", s, "worst", worst, flush=True)
