"""bench.py's heterogeneous sequence 5 on the CPU only: the reference (oracle/_ref, LAPACK dgesvd through MKL) next to our
restatement (oracle/port, Jacobi), frame by frame.  Run it on two different hosts: the frame at which the two part ways is
the same (an ill-conditioned 6x6 solve in Minimizer_RV's init phase), but WHICH of the two results the reference lands on
depends on the host CPU (MKL picks its code path by CPU model)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from rebvo_amd import edgehip, synth
from bench import tri

w, h, S, PF, Wm, K = 752, 480, 6, 12, 24, 40
p = edgehip.euroc_params(w, h)
intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
s = 5
scene = [f for f, _, _ in synth.billboard_sequence(w, h, PF, seed=101 + 7 * (s % S), traj_seed=29 + (s % S), **intr)]
ph = (s // S) % (2 * (PF - 1))
a = oracle.Oracle("ref", oracle.euroc_params(w, h))
b = oracle.Oracle("port", oracle.euroc_params(w, h))
print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0])
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    _, na = a.process_frame(scene[tri(k + ph, PF)], 0.05 * k)
    _, nb = b.process_frame(scene[tri(k + ph, PF)], 0.05 * k)
    dv = np.max(np.abs(np.array(na.V[:]) - np.array(nb.V[:])))
    dw = np.max(np.abs(np.array(na.W[:]) - np.array(nb.W[:])))
    print(f"frame {k}: kn {na.kn}/{nb.kn} klm_num ref {na.klm_num} restatement {nb.klm_num}  dV {dv:.2e} dW {dw:.2e}")
