"""A whole replay against the reference (tests/test_pipeline_gpu.py::_run without the asserts): per frame |dV|, |dW| relative to the
step, and at the end how many depths differ by more than rtol 1e-5 among KeyLines with identical matches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from rebvo_amd import edgehip, synth
from oracle import oracle

w, h, n, nseq = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 2
frames = [f for f, _, _ in synth.billboard_sequence(w, h, n)]
orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=nseq, nslots=3)
for k, f in enumerate(frames):
    _, nr = orc.process_frame(f, 0.05 * k)
    eh.upload_rgb(eh.next_slot(), np.stack([f] * nseq))
    eh.process_frame(0.05 * k)
    ng = eh.read_nav()[0]
    if k == 0:
        continue
    Vr, Wr = np.array(nr.V[:]), np.array(nr.W[:])
    step = np.linalg.norm(Vr) + np.linalg.norm(Wr)
    print("frame %d kn %d/%d  |dV|/step %.2e  |dW|/step %.2e  klm %d/%d  Kp diff %.2e" % (
        k, ng.kn, nr.kn, np.abs(np.array(ng.V[:]) - Vr).max() / step, np.abs(np.array(ng.W[:]) - Wr).max() / step, ng.klm_num, nr.klm_num, abs(ng.Kp - nr.Kp)))
slot = eh.cur_slot()
kg, mask = eh.download_keylines(0, slot)
kr = orc.keylines(orc.cur_slot())
same = kg["m_id"] == kr["m_id"]
rel = np.abs(kg["rho"] - kr["rho"]) / np.maximum(np.abs(kr["rho"]), 1e-7)
bad = same & (rel > 1e-5)
print("same matches %.5f  depths outside rtol 1e-5: %d of %d  max rel %.3e  median rel %.3e" % (same.mean(), bad.sum(), same.sum(), rel[same].max(), np.median(rel[same])))
for i in np.nonzero(bad)[0][:8]:
    print("  ikl %d  rho %.9g vs %.9g  s_rho %.6g vs %.6g  m_id %d m_num %d/%d" % (i, kg["rho"][i], kr["rho"][i], kg["s_rho"][i], kr["s_rho"][i], kg["m_id"][i], kg["m_num"][i], kr["m_num"][i]))
eh.close()
