"""bench.py's host-side helpers (CPU): the numbers it reports next to `value` must mean what DESIGN.md says."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _rot_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


def test_triangle_order_walks_the_pool_back_and_forth():
    assert [bench.tri(k, 4) for k in range(8)] == [0, 1, 2, 3, 2, 1, 0, 1]


def test_pose_rmse_of_identical_and_of_offset_trajectories():
    rs = np.random.RandomState(0)
    cpu = {k: (rs.normal(size=3).cumsum(), _rot_z(0.01 * k), rs.normal(size=3), rs.normal(size=3)) for k in range(10)}
    gpu = [tuple(np.array(x) for x in cpu[k]) for k in range(10)]
    r = bench.pose_rmse({0: gpu}, {0: cpu}, "reference")
    assert r["frames"] == 10 and r["position"] == 0 and r["rotation_rad"] == 0 and r["V"] == 0 and r["W"] == 0
    # a constant 3-4-0 position offset and a constant 1e-3 rad rotation offset
    gpu2 = [(p + np.array([3e-3, 4e-3, 0]), R @ _rot_z(1e-3), v, w) for p, R, v, w in gpu]
    r2 = bench.pose_rmse({0: gpu2}, {0: cpu}, "reference")
    assert abs(r2["position"] - 5e-3) < 1e-12 and abs(r2["rotation_rad"] - 1e-3) < 1e-9
    assert r2["path_length"] > 0 and abs(r2["position_rel"] - 5e-3 / r2["path_length"]) < 1e-12
    assert bench.pose_rmse(None, {0: cpu}, "reference") is None


def test_algorithmic_bytes_follow_the_survey_formulas():
    kn, n, r, B = 12000, 752 * 480, 40, 1024
    assert bench.algorithmic_bytes("B.try_velrot", kn, n, r, B) == 84 * kn * B
    assert bench.algorithmic_bytes("B.build_field", kn, n, r, B) == (4 * n + 4 * 2 * r * kn) * B
    # FordwardMatch's copy of the ten fields (100 B) is priced where it happens: inside k_directed when matching runs in one pass
    both = (4 * 40 + 2 * 168) * kn + (4 + 8 + 8 + 4 + 100) * kn
    for one_pass in (True, False):
        bench.ONE_PASS_MATCHING = one_pass
        d, f = bench.algorithmic_bytes("C.directed_matching", kn, n, r, 1), bench.algorithmic_bytes("C.forward_match", kn, n, r, 1)
        rot = bench.algorithmic_bytes("C.rotate", kn, n, r, 1) - 64 * kn      # one pass: the arbitration rides on rotate_keylines' pass
        assert d + (rot if one_pass else f) == both and d == (4 * 40 + 2 * 168 + (100 if one_pass else 0)) * kn
    bench.ONE_PASS_MATCHING = os.environ.get("EDGEHIP_FUSE_MATCH", "1") != "0"
    assert bench.algorithmic_bytes("no.such.group", kn, n, r, B) == 0


def test_pmc_traffic_comes_from_the_committed_counter_passes():
    js = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    nseq = js["_nseq"]
    t = bench.pmc_traffic("B.try_velrot", nseq)
    alg = bench.algorithmic_bytes("B.try_velrot", 12000, 752 * 480, 40, nseq)
    assert isinstance(t, int) and alg < t < 3 * alg        # measured traffic above, but within 3x of, the algorithmic bytes
    assert bench.pmc_traffic("B.try_velrot", nseq + 1) is None      # counters taken at another batch size are not used
    assert bench.pmc_traffic("B.lm_step_no_such", nseq) is None


def test_every_profiled_group_names_its_kernels():
    # every group bench.py may report as dominant has algorithmic bytes and a kernel list for the PMC lookup
    for g in ("A.detect", "A.level", "A.compact", "B.try_velrot", "B.build_field", "C.directed_matching", "C.forward_match"):
        assert bench.algorithmic_bytes(g, 1000, 1000, 40, 1) > 0 and bench.GROUP_KERNELS[g]


def test_stage_a_bytes_split_between_detector_and_join_add_up_to_the_survey_figure():
    """stage A = 3N + 4N + 168 kn.  Since round 3 the detector writes 24 B of a KeyLine and the join kernel the
    other 144 B (after reading the 24 back and probing three mask neighbours): the two groups' algorithmic bytes must add up to
    the survey's figure plus exactly those re-reads."""
    kn, n = 13000, 752 * 480
    a = bench.algorithmic_bytes("A.fused", kn, n, 40, 1) + bench.algorithmic_bytes("A.join_retune", kn, n, 40, 1)
    assert a == 3 * n + 4 * n + 168 * kn + (24 + 3 * 4) * kn
    assert bench.algorithmic_bytes("B.tvr_prepare", kn, n, 40, 1) == 0      # no stream left in it: latency, not priced


def test_committed_counters_carry_the_keyline_count_they_were_taken_at():
    js = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    assert js["_kn"] == bench.pmc_kn() and 5000 < js["_kn"] < 20000
    # a KeyLine-proportional kernel's traffic is compared with the algorithmic bytes at that count
    nseq, calib = js["_nseq"], bench.fetch_calibration()[0]
    tr, _ = bench.calibrated_traffic("B.try_velrot", nseq, js["_kn"], calib)
    ab = bench.algorithmic_bytes("B.try_velrot", js["_kn"], 752 * 480, 40, nseq)
    assert 0.9 < tr / ab < 1.6


def test_algorithmic_bytes_of_launches_that_carry_two_evaluations():
    kn, n, r, B = 12000, 752 * 480, 40, 1024
    # 12 evaluations in 9 launches (the two initialisation chains of Minimizer_RV share theirs)
    assert bench.algorithmic_bytes("B.try_velrot", kn, n, r, B, 12 / 9) == 84 * kn * B * 12 / 9
    assert bench.algorithmic_bytes("C.rescale", kn, n, r, 1) == 32 * kn     # one pass: the kernel keeps the KeyLines on chip


def test_wide_parity_counts_departures_and_attributes_them():
    """bench.py's free-running parity of many sequences: fed the reference's own records it reports nothing; a sequence whose
    velocity is off by more than the per-frame tolerance from some frame on is reported with that frame, and as outside
    tolerance at the last frame."""
    import pytest
    from oracle import oracle
    from rebvo_amd import edgehip, synth
    if not oracle.available("ref"):
        pytest.skip("oracle/_ref not built")
    w, h, n = 188, 120, 5
    frames = np.stack([f for f, _, _ in synth.billboard_sequence(w, h, 6)])
    idx_of = lambda s: [bench.tri(k + s, 6) for k in range(n)]
    log = np.zeros((n, 3), dtype=edgehip.NAV_DTYPE)
    for s in range(3):
        orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
        for k, i in enumerate(idx_of(s)):
            _, nav = orc.process_frame(frames[i], 0.05 * k)
            log[k, s]["V"], log[k, s]["W"], log[k, s]["Pos"], log[k, s]["kn"] = nav.V[:], nav.W[:], nav.Pos[:], nav.kn
        orc.close()
    summary, trajs = bench.wide_parity(log, [0, 1, 2], idx_of, frames, ("euroc", w, h, 0.05), 2, 2)
    assert summary["sequences_checked"] == 3 and summary["departures"] == [] and summary["sequences_outside_tolerance_at_last_frame"] == 0
    assert sorted(trajs) == [0, 1, 2] and sorted(trajs[1]) == [0, 1, 2]
    assert np.array_equal(trajs[2][2][2], log[4, 2]["V"])
    bad = log.copy()
    bad["V"][3:, 1, 0] += 1e-3
    summary, _ = bench.wide_parity(bad, [0, 1, 2], idx_of, frames, ("euroc", w, h, 0.05), 2, 2)
    assert [d["sequence"] for d in summary["departures"]] == [1]
    d = summary["departures"][0]
    assert d["first_frame_outside_tolerance"] == 3 and d["outside_tolerance_at_last_frame"] and d["knife_edge_frame"] in (True, False)
    assert summary["sequences_outside_tolerance_at_last_frame"] == 1
