#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE oracle (oracle/_ref, built in place from /root/reference).

Run in the build container (needs /root/reference):  python tools/make_golden.py
The fixtures are small (< 1 MB): per-frame scalars and hashes for a short sequence plus the complete
KeyLine list / mask of the last frame, so that tests can pin (a) the CPU restatement oracle/port and
(b) the HIP path even where oracle/_ref is not available.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from rebvo_amd import synth  # noqa: E402


KF_REQUESTS = [((0, 0, 0, 0, 0, 0), 1.0), ((0.003, -0.002, 0.001, 0.001, 0.002, -0.001), 1.1)]
KF_ARGS = (5.0, 30.0 * np.pi / 180.0, 5.0, 5, 2.0, 0)   # match_mod, match_ang, rho_tol, iter_max, reweight_distance, match_num_thresh


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make(name, w, h, frames, **over):
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h, **over))
    rec = dict(kn=[], tresh=[], retuned=[], mask_sha=[], dog_sha=[], img0_sha=[], V=[], W=[], Pos=[], klm_num=[],
               klm_fwd=[], s_rho_q=[], Kp=[], RKp=[], ok=[], score=[])
    prev = -1
    op = orc.p
    knife = []      # (frame, old KeyLine) pairs whose first TryVelRot evaluation the reference's own rounding noise decides
    for k, f in enumerate(frames):
        if k == len(frames) - 1:
            prev = orc.cur_slot()
        old = orc.keylines(orc.cur_slot()).copy() if k else None
        _, nav = orc.process_frame(f, 0.05 * k)
        s = orc.cur_slot()
        if k:
            knife += [(k, i) for i, _ in oracle.half_pixel_keylines(old, orc.field(s)[:, :, 1], op.ppx, op.ppy, nav.s_rho_q, op.w, op.h)]
        rec["kn"].append(nav.kn); rec["tresh"].append(nav.tresh); rec["retuned"].append(nav.retuned_thresh)
        rec["mask_sha"].append(sha(orc.mask(s))); rec["dog_sha"].append(sha(orc.plane(s, "dog")))
        rec["img0_sha"].append(sha(orc.plane(s, "img0")))
        rec["V"].append(nav.V[:]); rec["W"].append(nav.W[:]); rec["Pos"].append(nav.Pos[:])
        rec["klm_num"].append(nav.klm_num); rec["klm_fwd"].append(nav.klm_fwd); rec["s_rho_q"].append(nav.s_rho_q)
        rec["Kp"].append(nav.Kp); rec["RKp"].append(nav.RKp); rec["ok"].append(nav.estimation_ok)
        rec["score"].append(nav.score)
    s = orc.cur_slot()
    kl = orc.keylines(s)
    out = {k: np.array(v) for k, v in rec.items()}
    out["last_keylines"] = np.frombuffer(kl.tobytes(), dtype=np.uint8)
    out["last_mask"] = orc.mask(s).astype(np.int32)
    out["frames"] = np.stack([f[:, :, 0] for f in frames]).astype(np.uint8)  # r=g=b
    out["over"] = np.array(repr(sorted(over.items())))
    out["knife_edge"] = np.array(knife, dtype=np.int32).reshape(-1, 2)
    # key-frame tracker (kfvo::Minimizer_RV_KF): the previous frame's KeyLines against the field of the last
    # frame's, with the arguments kfvo::OptimizePosGT passes (kfvo.cpp:74) and two start poses
    kf = dict(X=[], RRV=[], ratio=[], mnum=[], mid_sha=[])
    for X0, Kr in KF_REQUESTS:
        r = orc.minimizer_rv_kf(s, prev, X0, Kr, float(rec["s_rho_q"][-1]), *KF_ARGS)
        kf["X"].append(r["X"]); kf["RRV"].append(r["RRV"]); kf["ratio"].append(r["score_ratio"]); kf["mnum"].append(r["mnum"])
        kf["mid_sha"].append(sha(orc.keylines(prev)["m_id_f"]))
    for k2, v in kf.items():
        out["kf_" + k2] = np.array(v)
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "kn", rec["kn"], "klm", rec["klm_num"], "knife-edge (frame, KeyLine):", knife, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if not oracle.available("ref"):
        raise SystemExit("oracle/_ref/libreforacle.so missing: run `make -C oracle ref` where /root/reference exists")
    make("rects_192x144", 192, 144, list(synth.rects_sequence(192, 144, 6, seed=3)))
    make("billboard_256x192", 256, 192, [f for f, _, _ in synth.billboard_sequence(256, 192, 7)],
         max_points=4000, reference_points=3000, track_points=3000, global_match_threshold=200)
    make("truncate_200x150", 200, 150, list(synth.rects_sequence(200, 150, 3, seed=5)),
         max_points=300, reference_points=250, track_points=250, global_match_threshold=50)
