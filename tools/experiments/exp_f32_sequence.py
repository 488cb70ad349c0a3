"""Per-frame distance between the device's float tracker and the reference's float instantiation over whole sequences, next to the
distance between the reference's own float and double instantiations (the yardstick for 'float accuracy')."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rebvo_amd import edgehip, synth
from oracle import oracle
w, h, n, B = 376, 240, 24, 3
frames = [f for f, _, _ in synth.billboard_sequence(w, h, n + B)]
eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=B, nslots=3)
eh.set_tracker_precision(32)
eh.set_nav_log(n)
for k in range(n):
    eh.upload_rgb(eh.next_slot(), np.stack([frames[k + s] for s in range(B)]))
    eh.process_frame(np.full(B, 0.05 * k))
log = eh.read_nav_log_array(0, n)
eh.close()
for s in range(B):
    o32, o64 = oracle.Oracle("ref", oracle.euroc_params(w, h)), oracle.Oracle("ref", oracle.euroc_params(w, h))
    o32.set_tracker_f32(1)
    for k in range(n):
        _, a = o32.process_frame(frames[k + s], 0.05 * k)
        _, b = o64.process_frame(frames[k + s], 0.05 * k)
        if k == 0: continue
        g = log[k, s]
        Xg, Xa, Xb = np.r_[g["V"], g["W"]], np.r_[np.array(a.V[:]), np.array(a.W[:])], np.r_[np.array(b.V[:]), np.array(b.W[:])]
        nr = np.linalg.norm(Xa)
        print(s, k, "dev32-ref32 %.2e   ref32-ref64 %.2e   klm %d %d %d  score %.6g %.6g" % (np.max(np.abs(Xg - Xa)) / nr, np.max(np.abs(Xa - Xb)) / nr,
              g["klm_num"], a.klm_num, b.klm_num, g["score"], a.score))

print("---- teacher-forced (the reference's float state injected before every frame) ----")
from oracle import teacher
for s in range(B):
    o32 = oracle.Oracle("ref", oracle.euroc_params(w, h))
    o32.set_tracker_f32(1)
    e1 = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
    e1.set_tracker_precision(32)
    tf = teacher.teacher_forced_replay(e1, o32, lambda k, s=s: frames[k + s], n, tol_rel=2e-4, tol_abs=2e-7)
    e1.close(); o32.close()
    rel = [max(a, b) / (np.linalg.norm(r[2]) + np.linalg.norm(r[3]) + 1e-30) for a, b, r in zip(tf["dV"], tf["dW"], tf["ref"])]
    print(s, "max rel", max(rel[1:]), "outside", [(o["frame"], o.get("dV"), o["kn_ok_klm"]) for o in tf["outside_tolerance"]])
    print("   per frame", " ".join("%.1e" % r for r in rel[1:]))
