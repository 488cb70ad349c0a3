"""Restart / knife-edge parity (GPU).

(1) Teacher-forced replay of bench.py's heterogeneous sequence 5 — the sequence whose free-running trajectory left the
    reference in BENCH_r02: with the reference's state injected before every frame, every frame (the knife-edge frame 8
    included) must agree with the reference inside the per-frame tolerance.
(2) Free-running, the device may leave the reference only on a frame the reference's own arithmetic leaves undecided
    (oracle.half_pixel_keylines; tests/test_knife_edge_cpu.py shows the reference doing the same against our CPU restatement).
(3) The 6x6 solves of Minimizer_RV on the device against TooN::SVD<>::backsub / TooN::Cholesky<6>::backsub themselves, on
    crafted systems either side of SVD's condition_no = 1e9 cut-off (TooN/SVD.h:37, 179; global_tracker.cpp:660-661, 711-712).
"""
import numpy as np
import pytest

from helpers import hetero_sequence, require_ref
from rebvo_amd import edgehip

pytestmark = pytest.mark.gpu

W, H, SEQ, NF = 752, 480, 5, 12


def _replay(forced):
    oracle = require_ref()
    from oracle import teacher
    orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=1, nslots=3)
    try:
        return teacher.teacher_forced_replay(eh, orc, hetero_sequence(SEQ), NF, forced=forced)
    finally:
        eh.close()
        orc.close()


def test_teacher_forced_replay_agrees_on_every_frame_including_the_knife_edge():
    r = _replay(True)
    assert 8 in [f["frame"] for f in r["knife_edge_frames"]]      # the frame of BENCH_r02's split is in the replay
    assert r["outside_tolerance"] == [], r["outside_tolerance"]
    assert r["max_dV"] < 1e-9 and r["max_dW"] < 1e-9


def test_free_running_replay_leaves_the_reference_only_at_a_knife_edge_frame():
    r = _replay(False)
    if r["outside_tolerance"]:
        first = r["outside_tolerance"][0]["frame"]
        assert first in [f["frame"] for f in r["knife_edge_frames"]], (first, r["knife_edge_frames"])
    # before the first knife-edge frame nothing may differ at all
    k0 = min(f["frame"] for f in r["knife_edge_frames"])
    assert max(r["dV"][:k0]) < 1e-9 and max(r["dW"][:k0]) < 1e-9


def _spd(rng, eig):
    q, _ = np.linalg.qr(rng.standard_normal((6, 6)))
    a = (q * np.asarray(eig)) @ q.T
    return (a + a.T) / 2


def test_lm_solve_against_toon_either_side_of_the_svd_cutoff():
    oracle = require_ref()
    orc = oracle.Oracle("ref", oracle.euroc_params(64, 48))
    eh = edgehip.EdgeHip(edgehip.euroc_params(64, 48), nseq=1, nslots=3)
    rng = np.random.default_rng(5)
    cases = []
    # what the tracker produces: JtJ + u I, condition 10..1e5
    for c in (1e1, 1e3, 1e5):
        cases.append(_spd(rng, 3e11 * np.array([1, 0.5, 0.1, 0.03, 3.0 / c, 1.0 / c])))
    # one and two singular values either side of s_max / 1e9: kept (ratio 0.5e9, 0.999e9), dropped (1.001e9, 2e9, 1e12)
    for ratio in (0.5e9, 0.999e9, 1.001e9, 2e9, 1e12):
        cases.append(_spd(rng, [1.0, 0.4, 0.2, 0.1, 0.05, 1.0 / ratio]))
        cases.append(_spd(rng, [7e10, 3e10, 2e9, 1e9, 7e10 / ratio, 0.3 * 7e10 / ratio]))
    # exactly singular and rank 3 (no data in some direction: a scene cut with a handful of matches)
    cases.append(_spd(rng, [1.0, 0.5, 0.2, 0.1, 0.05, 0.0]))
    cases.append(_spd(rng, [1.0, 0.5, 0.2, 0.0, 0.0, 0.0]))
    A = np.stack(cases)
    b = rng.standard_normal((len(A), 6)) * np.abs(A).max(axis=(1, 2))[:, None]
    h_svd = eh.lm_solve(A, b, svd_rule=True)
    h_chol = eh.lm_solve(A[:3], b[:3], svd_rule=False)
    for i in range(len(A)):
        want = orc.svd_backsub(A[i], b[i])
        scale = np.max(np.abs(want)) + 1e-300
        # a dropped direction contributes |b| / s_min ~ 1e9 times the rest when kept: agreement to 1e-5 of the solution's size
        # means both sides made the same keep / drop choice for every singular value
        assert np.max(np.abs(h_svd[i] - want)) <= 1e-5 * scale, (i, h_svd[i], want)
    for i in range(3):
        want = orc.svd_backsub(A[i], b[i], chol=True)
        assert np.max(np.abs(h_chol[i] - want)) <= 1e-9 * (np.max(np.abs(want)) + 1e-300), i
    eh.close()
    orc.close()
