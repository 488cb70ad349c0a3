"""How close is one TryVelRot evaluation to the reference's?  Prints max relative errors of F, J^T J, J^T F and whether the
residual memory is bit-identical, for the three template variants at three states (the fixture of tests/test_stage_b_gpu.py).
Run on the GPU box with the library under test in rebvo_amd/lib (tools/experiments/CALLS.md: r04_s)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from rebvo_amd import edgehip
from helpers import inject_pair, oracle_pair, rel_err

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (752, 480)
orc, so, sn, nav, frames = oracle_pair(w, h, 4)
eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=2)
inject_pair(eh, orc, so, sn)
orc.build_field(sn, 40, orc.retuned(sn))
eh.build_field(1, 40, -1.0)
s_rho_q = orc.quantile(so)
rs = np.random.RandomState(1)
worst = {}
for reweight, procjf in [(False, True), (True, True), (False, False)]:
    for X in (np.zeros(6), np.r_[np.array(nav.V[:]), np.array(nav.W[:])], rs.normal(size=6) * np.array([3e-3] * 3 + [2e-3] * 3)):
        X = np.asarray(X, np.float64)
        F0, _, _, r0 = orc.try_velrot(sn, so, X * 0.5, False, True, 0.5, s_rho_q, 0, 2.0)
        eh.try_velrot(1, 0, X * 0.5, False, True, 0.5, s_rho_q, 0, 2.0, resid_in=-1, resid_out=1)
        F, JtJ, JtF, r1 = orc.try_velrot(sn, so, X, reweight, procjf, 0.5, s_rho_q, 0, 2.0, resid_in=r0)
        Fg, JtJg, JtFg = eh.try_velrot(1, 0, X, reweight, procjf, 0.5, s_rho_q, 0, 2.0, resid_in=1, resid_out=2)
        kl_ref = orc.keylines(so)
        kn = len(kl_ref)
        rg = eh.download_resid(2)[0, :kn]
        used = ~(kl_ref["s_rho"] > s_rho_q)
        e = dict(F=rel_err(Fg[0], F), res_equal=bool(np.array_equal(rg[used], r1[used])), res=rel_err(rg[used], r1[used]))
        if procjf:
            e["JtJ"] = rel_err(JtJg[0], JtJ); e["JtF"] = rel_err(JtFg[0], JtF)
        print("reweight=%d procjf=%d |X|=%.2e kn=%d  " % (reweight, procjf, np.linalg.norm(X), kn) + "  ".join("%s=%s" % (k, ("%.2e" % v) if isinstance(v, float) else v) for k, v in e.items()))
        for k, v in e.items():
            if isinstance(v, float): worst[k] = max(worst.get(k, 0.0), v)
print("worst:", {k: "%.2e" % v for k, v in worst.items()})
eh.close()
