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
    assert bench.algorithmic_bytes("C.directed_matching", kn, n, r, 1) == (4 * 40 + 2 * 168) * kn
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
