"""bench.py's heterogeneous sequence 5 on the CPU: the reference with LAPACK (MKL dgesvd_), the reference with the
harness's own one-sided Jacobi behind the same dgesvd_ symbol, and our restatement — frame by frame, with the 6x6
systems of Minimizer_RV's init phase (global_tracker.cpp:659-661, 710-712) of the frame at which they part ways."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from rebvo_amd import edgehip, synth
from bench import tri

np.set_printoptions(linewidth=200, precision=17)
w, h, S, PF = 752, 480, 6, 12
s = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 10
p = edgehip.euroc_params(w, h)
intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
scene = [f for f, _, _ in synth.billboard_sequence(w, h, PF, seed=101 + 7 * (s % S), traj_seed=29 + (s % S), **intr)]
ph = (s // S) % (2 * (PF - 1))
print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(" \t:"))

runs = {}
for name, kind, backend in (("ref/lapack", "ref", 0), ("ref/jacobi", "ref", 1), ("port", "port", None)):
    o = oracle.Oracle(kind, oracle.euroc_params(w, h))
    if backend is not None:
        o.svd_backend(backend)
    navs, traces, ambs = [], [], []
    for k in range(nfr):
        old = o.keylines(o.cur_slot()).copy() if k else None
        o.svd_trace_start()
        _, n = o.process_frame(scene[tri(k + ph, PF)], 0.05 * k)
        traces.append(o.svd_trace_stop())
        navs.append((np.array(n.V[:]), np.array(n.W[:]), n.klm_num, n.score))
        amb = oracle.half_pixel_keylines(old, o.field(o.cur_slot())[:, :, 1], o.p.ppx, o.p.ppy, n.s_rho_q, w, h) if k else []
        ambs.append([(i, v, [float(x) for x in old["c_p"][i]], float(old["rho"][i]).hex()) for i, v in amb])
    runs[name] = (navs, traces, ambs)
    if backend is not None:
        o.svd_backend(0)
    o.close()

base = "ref/lapack"
for k in range(nfr):
    line = f"frame {k}:"
    for name in ("ref/jacobi", "port"):
        dv = np.max(np.abs(runs[base][0][k][0] - runs[name][0][k][0]))
        dw = np.max(np.abs(runs[base][0][k][1] - runs[name][0][k][1]))
        line += f"  [{name}] dV {dv:.2e} dW {dw:.2e} klm {runs[name][0][k][2]}/{runs[base][0][k][2]}"
    print(line)
    if runs[base][2][k]:
        print(f"    knife-edge KeyLines of frame {k} (ikl, field entries at the candidate pixels, c_p, rho bits): "
              + "; ".join(f"{nm}: {runs[nm][2][k]}" for nm in runs))
    bad = max(np.max(np.abs(runs[base][0][k][0] - runs[n_][0][k][0])) for n_ in ("ref/jacobi", "port")) > 1e-9
    if bad:
        print(f"--- frame {k}: decompositions of the init phase, in call order ---")
        for name in runs:
            for j, r in enumerate(runs[name][1][k]):
                sv = np.sort(np.abs(r["s"]))[::-1]
                print(f"  {name:11s} solve {j}: s = {sv}  s0/s5 = {sv[0] / sv[5]:.6e}  dropped = {int(np.sum(sv * 1e9 <= sv[0]))}")
        for j in range(len(runs[base][1][k])):
            for name in ("ref/jacobi", "port"):
                if j < len(runs[name][1][k]):
                    dA = np.max(np.abs(runs[base][1][k][j]["A"] - runs[name][1][k][j]["A"]))
                    sc = np.max(np.abs(runs[base][1][k][j]["A"]))
                    print(f"  solve {j}: max|A({base}) - A({name})| = {dA:.3e}  (max|A| = {sc:.3e})")
        break
