"""GPU, BASELINE size (752x480): size-independent properties of the path that hold whatever the oracle says —
run-to-run determinism, batch invariance, structural invariants of the edge map (raster order, mask <-> id
consistency, neighbour links), MaxPoints truncation = prefix of the untruncated list, field idempotence, and
rescaling linearity."""
import numpy as np
import pytest

from rebvo_amd import edgehip, synth

pytestmark = pytest.mark.gpu
W, H = 752, 480


@pytest.fixture(scope="module")
def frames():
    return [f for f, _, _ in synth.billboard_sequence(W, H, 5)]


def _run(frames, nseq, order=None, **over):
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H, **over), nseq=nseq, nslots=3)
    navs = []
    for k in range(len(frames)):
        batch = np.stack([frames[(k + (order[s] if order else 0)) % len(frames)] for s in range(nseq)])
        eh.upload_rgb(eh.next_slot(), batch)
        eh.process_frame(0.05 * k)
        navs.append(eh.read_nav())
    return eh, navs


def _nav_tuple(n):
    return (n.kn, n.klm_num, n.klm_fwd, n.estimation_ok, tuple(n.V[:]), tuple(n.W[:]), tuple(n.Pos[:]), n.Kp, n.tresh, n.score)


def test_deterministic_and_batch_invariant(frames):
    """Same frames -> bit-identical records, run after run, alone or as any member of a batch."""
    eh1, a = _run(frames, 1)
    eh2, b = _run(frames, 1)
    eh3, c = _run(frames, 5, order=[0, 2, 0, 1, 0])          # sequences 0, 2, 4 see the same frames as the solo runs
    for k in range(len(frames)):
        assert _nav_tuple(a[k][0]) == _nav_tuple(b[k][0]), k
        for s in (0, 2, 4):
            assert _nav_tuple(a[k][0]) == _nav_tuple(c[k][s]), (k, s)
    k1, m1 = eh1.download_keylines(0, eh1.cur_slot())
    k3, m3 = eh3.download_keylines(4, eh3.cur_slot())
    assert np.array_equal(m1, m3) and k1.tobytes() == k3.tobytes()
    for e in (eh1, eh2, eh3):
        e.close()


def test_edge_map_structure(frames):
    eh, navs = _run(frames, 2, order=[0, 1])
    for s in range(2):
        kl, mask = eh.download_keylines(s, eh.cur_slot())
        kn = len(kl)
        assert kn == navs[-1][s].kn and 0 < kn <= 16000
        p = kl["p_inx"].astype(np.int64)
        assert np.all(np.diff(p) > 0), "KeyLine ids follow raster order"
        flat = mask.ravel()
        assert np.array_equal(flat[p], np.arange(kn)) and (flat >= 0).sum() == kn, "mask <-> id bijection"
        y, x = p // W, p % W
        assert x.min() >= 2 and x.max() < W - 2 and y.min() >= 2 and y.max() < H - 2           # scan window, edge_finder.cpp:105
        assert np.all(np.abs(kl["c_p"][:, 0] - x) <= 0.5) and np.all(np.abs(kl["c_p"][:, 1] - y) <= 0.5)   # sub-pixel offset
        assert np.allclose(np.linalg.norm(kl["u_m"], axis=1), 1.0, atol=1e-6)
        n_id, p_id = kl["n_id"], kl["p_id"]
        has = n_id >= 0
        assert has.mean() > 0.5
        # join_edges: the next KeyLine is one of the three neighbours in the tangent quadrant; p_id is the LAST writer
        assert np.all(np.abs(x[n_id[has]] - x[has]) <= 1) and np.all(np.abs(y[n_id[has]] - y[has]) <= 1)
        writers = np.full(kn, -1)
        np.maximum.at(writers, n_id[has], np.nonzero(has)[0])
        assert np.array_equal(writers, p_id)
        assert np.all((kl["rho"] >= 1e-3) & (kl["rho"] <= 20.0)) and np.all(kl["s_rho"] > 0)
        assert np.all(kl["m_id"][kl["m_num"] > 0] >= 0)
    eh.close()


def test_max_points_truncation_is_a_prefix(frames):
    """kl_max cuts the raster-ordered list (edge_finder.cpp:203-209): with a fixed threshold the truncated list is the
    first MaxPoints KeyLines of the untruncated one, and the mask beyond the cut is empty."""
    fixed = dict(auto_gain=0.0, detector_thresh=0.02)
    eh_full = edgehip.EdgeHip(edgehip.euroc_params(W, H, **fixed), nseq=1, nslots=2)
    eh_cut = edgehip.EdgeHip(edgehip.euroc_params(W, H, max_points=3000, **fixed), nseq=1, nslots=2)
    for eh in (eh_full, eh_cut):
        eh.upload_rgb(0, frames[0])
        eh.stage_a(0)
    kf, mf = eh_full.download_keylines(0, 0)
    kc, mc = eh_cut.download_keylines(0, 0)
    assert len(kf) > 3000 and len(kc) == 3000
    for fld in ("p_inx", "m_m", "u_m", "n_m", "c_p", "p_m"):
        assert np.array_equal(kc[fld], kf[fld][:3000]), fld
    last = kc["p_inx"][-1]
    assert np.array_equal(mc.ravel()[:last + 1], np.where(mf.ravel()[:last + 1] < 3000, mf.ravel()[:last + 1], -1))
    assert np.all(mc.ravel()[last + 1:] == -1)
    eh_full.close()
    eh_cut.close()


def test_field_idempotent_and_consistent(frames):
    # debug_planes: keep the field's distances on the device (the tracker only gathers the KeyLine-index plane)
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H, debug_planes=1), nseq=1, nslots=2)
    eh.upload_rgb(0, frames[0])
    eh.stage_a(0)
    eh.build_field(0, 40, -1.0)
    f1 = eh.download_field(0)
    eh.build_field(0, 40, -1.0)
    f2 = eh.download_field(0)
    assert np.array_equal(f1, f2)
    kl, _ = eh.download_keylines(0, 0)
    ikl, dist = f1[..., 1], f1[..., 0]
    hit = ikl >= 0
    assert hit.any() and ikl.max() < len(kl) and dist[hit].max() <= 40
    # a KeyLine strong enough for the field owns its own pixel at distance 0 (t = 0 sample), unless a later one shares it
    thr = eh.get_state(0).retuned_thresh
    strong = np.nonzero(kl["n_m"] >= thr)[0]
    cx, cy = np.round(kl["c_p"][strong, 0]).astype(int), np.round(kl["c_p"][strong, 1]).astype(int)
    assert np.all(dist[cy, cx] == 0) and np.all(ikl[cy, cx] >= strong)
    eh.close()


def test_rescale_is_linear(frames):
    """EstimateReScalingOpt with DoReScaling: rho and s_rho are divided by the returned Kp (edge_tracker.cpp:1133-1138),
    so running the same frames with and without rescaling differs by exactly that factor on the last edge map."""
    eh0, n0 = _run(frames[:3], 1)
    eh1, n1 = _run(frames[:3], 1, do_rescaling=1)
    # identical up to (and including) the tracker of frame 2 only if frame 1's rescale was ~1; compare frame 1 -> exact factor
    eha, _ = _run(frames[:2], 1)
    ehb, nb = _run(frames[:2], 1, do_rescaling=1)
    ka, _ = eha.download_keylines(0, eha.cur_slot())
    kb, _ = ehb.download_keylines(0, ehb.cur_slot())
    Kp = nb[-1][0].Kp
    assert Kp > 0 and np.isfinite(Kp)
    assert np.allclose(kb["rho"] * Kp, ka["rho"], rtol=1e-14, atol=0)
    assert np.allclose(kb["s_rho"] * Kp, ka["s_rho"], rtol=1e-14, atol=0)
    for e in (eh0, eh1, eha, ehb):
        e.close()
