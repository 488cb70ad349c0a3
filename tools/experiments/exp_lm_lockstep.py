"""Lock-step Minimizer_RV over frame 8 of the pool-6 replay: every TryVelRot evaluation on the reference and on the GPU
with the same X and the same residual history."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
from rebvo_amd import edgehip, synth
from oracle import oracle
from helpers import inject_pair, rel_err
w, h, npool = 752, 480, 6
NW = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
print("field ikl equal", np.array_equal(fr[..., 1], fg[..., 1]), "dist equal", np.array_equal(fr[..., 0][fr[..., 1] >= 0], fg[..., 0][fr[..., 1] >= 0]))
s_rho_q = orc.quantile(so); eh.quantile(0)
print("s_rho_q", s_rho_q, eh.get_state(0).s_rho_q, "kn old", orc.kn(so), "new", orc.kn(sn))
kn = orc.kn(so)
R = {"Z": np.zeros(kn), "T": np.zeros(kn), "N": np.zeros(kn)}
G = {"Z": -1, "T": 2, "N": 1, "X": 0}
n_eval = [0]
def ev(X, rw, jf, rin, rout):
    F, JtJ, JtF, r = orc.try_velrot(sn, so, X, rw, jf, 0.5, s_rho_q, 0, 2.0, resid_in=R[rin])
    R[rout] = r
    Fg, JtJg, JtFg = eh.try_velrot(1, 0, X, rw, jf, 0.5, s_rho_q, 0, 2.0, resid_in=G[rin], resid_out=G[rout])
    rg = eh.download_resid(G[rout])[0, :kn]
    kl = orc.keylines(so)
    skipped = kl["s_rho"] > s_rho_q
    dres = np.abs(rg[~skipped] - r[~skipped]).max()
    klg, _ = eh.download_keylines(0, 0, want_mask=False)
    nmid = int((kl["m_id_f"] != klg["m_id_f"]).sum())
    n_eval[0] += 1
    print("eval", n_eval[0], "rw", int(rw), "jf", int(jf), "F %.10e %.10e rel %.1e" % (F, Fg[0], rel_err(Fg[0], F)),
          "JtJ %.1e JtF %.1e" % ((rel_err(JtJg[0], JtJ), rel_err(JtFg[0], JtF)) if jf else (0, 0)), "dres %.1e" % dres, "m_id_f diff", nmid, flush=True)
    return F, JtJ, JtF
def init_phase(X, out):
    F, JtJ, JtF = ev(X, False, True, "Z", out)
    u, v = 1e-3 * JtJ.max(), 2.0
    for i in range(2):
        hh = np.linalg.solve(JtJ + np.eye(6) * u, -JtF)
        Xn = X + hh
        if i == 1:
            Fn, Jn, Jfn = ev(Xn, False, False, "Z", out); gain = F - Fn
        else:
            Fn, Jn, Jfn = ev(Xn, False, True, "Z", out); gain = (F - Fn) / (0.5 * hh @ (u * hh - JtF))
        if gain > 0:
            F, X = Fn, Xn
            if i == 0: JtJ, JtF = Jn, Jfn
            u *= max(0.33, 1 - (2 * gain - 1) ** 3); v = 2.0
        else:
            u *= v; v *= 2
    return X, F
Xa, Fa = init_phase(np.zeros(6), "T")
Xb, Fb = init_phase(np.r_[np.array(nav.V[:]), np.array(nav.W[:])], "N")
print("init: zero F", Fa, "prior F", Fb)
if Fb > Fa:
    X, cur = Xa, "T"; new = "N"
else:
    X, cur = Xb, "N"; new = "T"
F, JtJ, JtF = ev(X, True, True, cur, new)
u, v = 1e-3 * JtJ.max(), 2.0
for it in range(5):
    hh = np.linalg.solve(JtJ + np.eye(6) * u, -JtF)
    Xn = X + hh
    Fn, Jn, Jfn = ev(Xn, True, True, cur, new)
    gain = (F - Fn) / (0.5 * hh @ (u * hh - JtF))
    print("   gain", gain)
    if gain > 0:
        F, X, JtJ, JtF = Fn, Xn, Jn, Jfn
        u *= max(0.33, 1 - (2 * gain - 1) ** 3); v = 2.0
        cur, new = new, cur
    else:
        u *= v; v *= 2
print("final X", X, "F", F)
