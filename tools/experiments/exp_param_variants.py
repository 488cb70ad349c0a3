"""Whole-path parity under parameter sets the shipped configs do not use."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from rebvo_amd import edgehip, synth
from oracle import oracle
w, h, nf = 376, 240, 14
frames = [f for f, _, _ in synth.billboard_sequence(w, h, nf, seed=21)]
VARIANTS = [dict(tracker_init_type=0), dict(tracker_init_type=1), dict(tracker_init_iter_num=3), dict(tracker_init_iter_num=1),
            dict(do_rescaling=1), dict(search_range=12), dict(search_range=64), dict(match_num_thresh=2), dict(match_num_thresh=6),
            dict(pos_neg_thresh=0.2), dict(dog_thresh=0.2), dict(reweight_distance=1.0), dict(tracker_match_thresh=1.0),
            dict(regularize_thresh=0.2), dict(max_points=3000, reference_points=2500), dict(global_match_threshold=20000),
            dict(tracker_iter_num=1), dict(tracker_iter_num=12), dict(match_thresh_angle=20.0, match_thresh_module=0.3),
            dict(loc_unc_match=1.0, loc_unc=2.0), dict(reshape_q_abs=1e-2, reshape_q_rel=1e-2), dict(qcut_quantile=0.5),
            dict(qcut_nbins=50), dict(auto_gain=5e-6), dict(detector_thresh=0.05, auto_gain=0.0), dict(track_points=2000)]
for over in VARIANTS:
    try:
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h, **over), nseq=1, nslots=3)
    except Exception as e:
        print(over, "create failed:", str(e)[:100]); continue
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h, **over))
    worst, first, oks = 0.0, None, 0
    for k in range(nf):
        eh.upload_rgb(eh.next_slot(), frames[k][None]); eh.process_frame(0.05 * k)
        ng = eh.read_nav()[0]
        _, nr = orc.process_frame(frames[k], 0.05 * k)
        if k == 0: continue
        d = max(np.abs(np.array(ng.V[:]) - np.array(nr.V[:])).max(), np.abs(np.array(ng.W[:]) - np.array(nr.W[:])).max(),
                np.abs(np.array(ng.Pos[:]) - np.array(nr.Pos[:])).max(), abs(ng.Kp - nr.Kp))
        if (ng.kn, ng.estimation_ok, ng.klm_num) != (nr.kn, nr.estimation_ok, nr.klm_num): d = max(d, 1.0)
        if d > 1e-9 and first is None: first = (k, d, (ng.kn, ng.estimation_ok, ng.klm_num), (nr.kn, nr.estimation_ok, nr.klm_num))
        worst = max(worst, d); oks += nr.estimation_ok
    print(over, "worst %.2e" % worst, "first", first, "ref ok frames", oks, flush=True)
    eh.close()
