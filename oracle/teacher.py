"""Teacher-forced replay and knife-edge attribution (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

Why.  The path is a chain of discrete decisions (which pixel a re-projected KeyLine rounds to, which KeyLine matches, which
LM step is accepted) fed by fp64 sums.  Two correct implementations agree on the sums to ~1e-15, so they agree on every
decision — except where the *reference's own* decision hangs on the last bit of its input.  One such place is built into
the algorithm: `oracle.half_pixel_keylines` (KeyLines detected exactly on a half pixel, evaluated at X = 0 by the
zero-initialised trial of Minimizer_RV, global_tracker.cpp:650-651).  On such a frame two runs of the reference itself —
with another LAPACK, on another CPU — may differ by one match, |dV| ~ 1e-6, and then drift apart for good.

A free-running comparison cannot tell a bug from such a frame; a teacher-forced one can.  Before every frame the path
under test receives the reference's state bit for bit (KeyLines of the previous edge map with their depths, velocity
prior, pose, threshold controller), runs ONE frame, and is compared with the reference's result for that frame.  The
projection at X = 0 is the same sequence of IEEE operations on both sides, so with identical input bits even a knife-edge
frame must agree; a frame outside tolerance under teacher forcing is a defect of the path under test, full stop.
"""
import numpy as np

from . import oracle as _o


def _edgehip_kl(kl):
    from rebvo_amd import edgehip
    return np.frombuffer(np.ascontiguousarray(kl).tobytes(), dtype=edgehip.KEYLINE_DTYPE).copy()


def inject_reference_state(eh, orc, seq=0):
    """Hand the device sequence `seq` the state the reference carries into its next frame."""
    so = orc.cur_slot()
    st = orc.seq_state()
    eh.upload_keylines(seq, eh.cur_slot(), _edgehip_kl(orc.keylines(so)), orc.mask(so), orc.retuned(so))
    g = eh.get_state(seq)
    g.tresh, g.l_kl_num = st.tresh, st.l_kl_num
    g.Kp, g.P_Kp, g.K, g.t_prev = st.Kp, st.P_Kp, st.K, st.t_prev
    for i in range(3):
        g.V[i], g.W[i], g.Pos[i] = st.V[i], st.W[i], st.Pos[i]
    for i in range(9):
        g.Pose[i] = st.Pose[i]
    g.retuned_thresh = orc.retuned(so)
    eh.set_state(seq, g)


def teacher_forced_replay(eh, orc, frame_of, nframes, dt=0.05, forced=True, tol_rel=1e-6, tol_abs=1e-9):
    """Replay `nframes` frames (frame_of(k) -> RGB24 array) on the reference `orc` and on sequence 0 of the device context
    `eh` (nseq = 1), frame by frame.  forced: inject the reference's state into the device before every frame.

    Returns a dict: per-frame |dV|, |dW| and count mismatches, the frames outside tolerance, the knife-edge frames of the
    reference's own trajectory (with the KeyLines that make them so) and the reference trajectory itself."""
    p = orc.p
    out = {"frames": nframes, "forced": bool(forced), "dV": [], "dW": [], "outside_tolerance": [], "knife_edge_frames": [],
           "ref": [], "dev": []}
    for k in range(nframes):
        img = frame_of(k)
        old = orc.keylines(orc.cur_slot()).copy() if k else None
        if forced and k:
            inject_reference_state(eh, orc, 0)
        eh.upload_rgb(eh.next_slot(), img[None])
        eh.process_frame(dt * k)
        ng = eh.read_nav()[0]
        _, nr = orc.process_frame(img, dt * k)
        out["ref"].append((np.array(nr.Pos[:]), np.array(nr.Pose[:]).reshape(3, 3), np.array(nr.V[:]), np.array(nr.W[:])))
        out["dev"].append((np.array(ng.Pos[:]), np.array(ng.Pose[:]).reshape(3, 3), np.array(ng.V[:]), np.array(ng.W[:])))
        if k == 0:
            out["dV"].append(0.0)
            out["dW"].append(0.0)
            if ng.kn != nr.kn:
                out["outside_tolerance"].append({"frame": 0, "kn": [int(ng.kn), int(nr.kn)]})
            continue
        amb = _o.half_pixel_keylines(old, orc.field(orc.cur_slot())[:, :, 1], p.ppx, p.ppy, nr.s_rho_q, p.w, p.h)
        if amb:
            out["knife_edge_frames"].append({"frame": k, "keylines": [{"ikl": i, "c_p": [float(x) for x in old["c_p"][i]],
                                                                       "field_at_candidate_pixels": v} for i, v in amb[:8]]})
        step = float(np.linalg.norm(nr.V[:]) + np.linalg.norm(nr.W[:]))
        fin = bool(np.all(np.isfinite(nr.V[:])) and np.all(np.isfinite(nr.W[:])))
        dv = float(np.max(np.abs(np.array(ng.V[:]) - np.array(nr.V[:])))) if fin else 0.0
        dw = float(np.max(np.abs(np.array(ng.W[:]) - np.array(nr.W[:])))) if fin else 0.0
        out["dV"].append(dv)
        out["dW"].append(dw)
        counts_dev = (int(ng.kn), int(ng.estimation_ok), int(ng.klm_num))
        counts_ref = (int(nr.kn), int(nr.estimation_ok), int(nr.klm_num))
        tol = tol_rel * step + tol_abs
        if counts_dev != counts_ref or dv > tol or dw > tol:
            out["outside_tolerance"].append({"frame": k, "dV": dv, "dW": dw, "tolerance": tol,
                                             "kn_ok_klm": [list(counts_dev), list(counts_ref)],
                                             "knife_edge": bool(amb)})
    out["max_dV"] = float(max(out["dV"]))
    out["max_dW"] = float(max(out["dW"]))
    return out
