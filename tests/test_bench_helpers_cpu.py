"""bench.py's host-side helpers (CPU): the numbers it reports next to `value` must mean what DESIGN.md says."""
import json
import time
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
    # the field builder is priced on what its tiled form moves (records, bin entries, the 2-byte index plane), not on the reference's
    # 8-byte scatter per sample, which stays as the survey-formula figure (the two differed 2x)
    tiles = 1 + 2 * r * (4 / np.pi) / 64
    assert bench.algorithmic_bytes("B.build_field", kn, n, r, B) == (2 * n + 44 * kn + 40 * kn * tiles) * B
    assert bench.survey_bytes("B.build_field", kn, n, r, B) == (8 * n + 8 * 2 * r * kn) * B
    assert bench.survey_bytes("B.build_field", kn, n, r, B) > 2 * bench.algorithmic_bytes("B.build_field", kn, n, r, B)
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
    t = bench.pmc_traffic("B.try_velrot", nseq, check_stamp=False)
    alg = bench.algorithmic_bytes("B.try_velrot", js.get("_kn", 12000), 752 * 480, 40, nseq)
    assert isinstance(t, int) and 0.8 * alg < t < 3 * alg        # measured traffic near, and within 3x of, the algorithmic bytes
    assert bench.pmc_traffic("B.try_velrot", nseq + 1, check_stamp=False) is None      # counters taken at another batch size are not used
    assert bench.pmc_traffic("B.lm_step_no_such", nseq, check_stamp=False) is None


def test_counters_of_other_sources_than_the_running_library_are_not_used(tmp_path, monkeypatch):
    """profiles/pmc_latest.json carries the sha of the library sources its passes were taken with (tools/gpu_round.sh); with any
    other sources in the tree `traffic` is null rather than a number of some other code."""
    js = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    nseq = js["_nseq"]
    sha = bench.library_source_sha()
    assert isinstance(sha, str) and len(sha) == 16
    for stamp, usable in ((sha, True), ("0" * 16, False), (None, False)):
        js2 = dict(js)
        js2["_src_sha"] = stamp
        (tmp_path / "profiles").mkdir(exist_ok=True)
        json.dump(js2, open(tmp_path / "profiles" / "pmc_latest.json", "w"))
        monkeypatch.setattr(bench, "PMC_FILE", str(tmp_path / "profiles" / "pmc_latest.json"))
        bench._PMC_CACHE.clear()
        assert (bench.pmc_traffic("B.try_velrot", nseq) is not None) == usable
        assert ("match" in bench.pmc_stamp_note() and "NO match" not in bench.pmc_stamp_note()) == usable
    bench._PMC_CACHE.clear()


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
    assert js["_kn"] == bench.pmc_kn(check_stamp=False) and 5000 < js["_kn"] < 20000
    # a KeyLine-proportional kernel's traffic is compared with the algorithmic bytes at that count
    nseq, calib = js["_nseq"], bench.fetch_calibration()[0]
    tr, _ = bench.calibrated_traffic("B.try_velrot", nseq, js["_kn"], calib, check_stamp=False)
    ab = bench.algorithmic_bytes("B.try_velrot", js["_kn"], 752 * 480, 40, nseq)
    assert 0.8 < tr / ab < 1.6


def test_algorithmic_bytes_of_launches_that_carry_two_evaluations():
    kn, n, r, B = 12000, 752 * 480, 40, 1024
    # a two-chain launch streams the KeyLine's own 40 B once and gathers 44 B per evaluation: that, not 2 x 84, is what it owes;
    # SURVEY 8(d)'s figure stays available as the survey formula
    assert bench.algorithmic_bytes("B.try_velrot2", kn, n, r, B) == (40 + 2 * 44) * kn * B
    assert bench.survey_bytes("B.try_velrot2", kn, n, r, B) == 2 * 84 * kn * B
    assert bench.survey_bytes("B.try_velrot", kn, n, r, B) == bench.algorithmic_bytes("B.try_velrot", kn, n, r, B) == 84 * kn * B
    assert bench.algorithmic_bytes("C.rescale", kn, n, r, 1) == 32 * kn     # one pass: the kernel keeps the KeyLines on chip
    assert bench.survey_bytes("C.rescale", kn, n, r, 1) == 5 * 32 * kn


def _canned_full_record():
    """A record of the size a default run produces (every optional object present, long notes, 20 kernel groups, 33 departures)."""
    groups = ["A.fused", "A.join_retune", "B.quantile", "B.build_field", "B.tvr_prepare", "B.try_velrot", "B.try_velrot2", "B.lm_step",
              "C.rotate", "C.directed_matching", "C.regularize_ekf", "C.rescale", "C.pose", "C.forward_match", "A.level", "A.compact"]
    dep = [{"sequence": i, "first_frame_outside_tolerance": 8, "knife_edge_frame": True, "outside_tolerance_at_last_frame": True,
            "position_error_at_last_frame": 0.006869748016065307, "knife_edge_frames_of_the_reference": list(range(12))} for i in range(33)]
    return {
        "metric": "frames/sec (DoG+extract+track+depth) 752x480 EuRoC", "value": 100213.7, "unit": "frames/s", "n_gpus": 1, "steps": 20,
        "warmup": 5, "ms_per_step": 10.2182, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 scale-space / f64 tracker+EKF", "data": "synthetic",
        "config": {"workload": "full path " + "x" * 600, "input": "every sequence reads its own RGB24 copy " + "y" * 300, "dataset": None,
                   "sequences_per_gpu": 1024, "contexts_per_gpu": 1, "sequences_per_launch": 1024, "stream_overlap": False,
                   "nav_gather": "ok", "nav_gather_info": {"backend": "rccl", "ranks": 1, "blocks": 4, "records": 20480, "equals_device_log": True},
                   "frames_per_step": 1024, "keylines_per_frame": 12345.6, "keylines_per_frame_timed_mean": 12345.6,
                   "tryvelrot_evals_per_frame": 12, "estimation_ok": "1024/1024", "algorithmic_MB_per_frame": 39.64, "whole_path_hbm_frac": 0.49655},
        "roofline": {"bound": "hbm", "kernel": "A.fused", "achieved": 1083.27, "peak": 8000.0, "unit": "GB/s", "frac": 0.13541,
                     "frac_on_traffic": 0.10833, "traffic": 2312345678, "launch_us": 2668.12, "algorithmic_bytes_per_launch": 2890123456,
                     "launches_timed": 20, "issue_frac": 0.7512345, "traffic_note": "z" * 400},
        "cpu_baseline": {"value": 49.41, "unit": "frames/s", "ms_per_frame": 20.237, "median_ms": 20.062, "p95_ms": 21.875, "frames": 100,
                         "cores": 1, "kind": "reference", "cpu_model": "AMD EPYC 9575F 64-Core Processor", "usable_cores": 128,
                         "sample": "s" * 500,
                         "modes": {"serial_1_core": {"value": 49.41, "unit": "frames/s", "ms_per_frame": 20.2, "cores": 1},
                                   "reference_threads_2_cores": {"value": 63.78, "unit": "frames/s", "cores": 2},
                                   "node_saturating": {"value": 1712.3, "unit": "frames/s", "processes": 42, "cores": 84, "sample": "t" * 300}}},
        "pose_rmse": {"position": 1.1163047506248056e-14, "rotation_rad": 2.220446049250313e-16, "V": 3.3e-15, "W": 1.1e-16, "frames": 660,
                      "sequences": list(range(0, 1024, 32)) + [1023], "position_per_sequence": {i: 1e-14 for i in range(33)},
                      "path_length": 12.3, "position_rel": 2.99e-14, "vs": "CPU reference " + "v" * 200,
                      "free_running_parity": {"sequences_checked": 33, "frames_per_sequence": 25, "reference_processes": 33,
                                              "sequences_outside_tolerance_at_last_frame": 1, "departures": dep,
                                              "departures_on_knife_edge_frames": 1, "departures_elsewhere": 0,
                                              "max_abs_dVW_while_inside_tolerance": 1.7019281105951078e-09, "tolerance": "w" * 200}},
        "kernel_us_per_step": {g: 1234.5 + i for i, g in enumerate(groups)},
        "kernel_us_per_step_source": "k" * 300,
        "roofline_kernels": {g: {"launch_us": 355.1, "achieved_GBs": 3391.2, "frac": 0.4239, "algorithmic_bytes_per_launch": 1204000000,
                                 "traffic": 1204000000, "frac_on_traffic": 0.42, "issue_frac": 0.5, "launches_per_step": 6} for g in groups},
        "scaling_measured": False, "traffic_source": "p" * 300, "traffic_calibration": {"factors_true_over_reported": {"stream": 2.0}},
        "batch_sweep": [{"sequences_per_launch": n, "frames_per_s": 3571.4, "ms_per_step": 0.28} for n in (1, 8, 64, 1024)],
        "single_sequence_ms_per_frame": 0.2801, "single_sequence_note": "n" * 300,
        "host_surface": {"single_camera_fps": 2612.3, "objects_8_fps": 14012.9, "objects_64_fps": 48211.0, "what": "h" * 70,
                         "detail": {"a": list(range(100))}},
        "heterogeneous": {"value": 92819.1, "free_running_parity": {"departures": dep}, "teacher_forced": {"per_sequence": {i: dep[0] for i in range(8)}}},
        "extras": {n: {"roofline": {"x": "q" * 500}, "cpu_baseline": {"sample": "r" * 500}} for n in ("stage_a", "tum_undistort", "imu")},
        "pcie_inclusive": {"rgb24": {"value": 61234.5}, "grey8": {"value": 98611.2}, "note": "m" * 300},
    }


def test_the_one_line_stays_under_4_kb_and_is_strict_json():
    """Round 4's line had grown to 22 KB and the driver could not parse it.  The line is now built by compact_line() from the full
    record: whatever the record holds, the line is one line of strict JSON below LINE_LIMIT with the contract's keys, the
    roofline, the CPU baseline with its three modes as numbers, and the pose check with its parity counts."""
    full = _canned_full_record()
    assert len(json.dumps(full)) > 20000      # the canned record is as large as round 4's line
    line = bench.compact_line(full)
    text = json.dumps(line, allow_nan=False)
    assert bench.LINE_LIMIT == 4096 and len(text) < bench.LINE_LIMIT and "\n" not in text
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "pose_rmse"):
        assert k in back, k
    assert back["value"] == 100213.7 and back["config"]["workload"].startswith("full path") and "model" not in back["config"]
    rf = back["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "A.fused" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] == 2312345678 and rf["frac_on_traffic"] and rf["issue_frac"] and rf["launch_us"] and rf["algorithmic_bytes_per_launch"]
    cb = back["cpu_baseline"]
    assert cb["value"] == 49.41 and cb["cores"] == 1 and cb["kind"] == "reference" and len(cb["sample"]) <= 200
    assert cb["modes"] == {"serial_1_core": 49.41, "reference_threads_2_cores": 63.78, "node_saturating": 1712.3}
    pr = back["pose_rmse"]
    assert pr["position"] > 0 and pr["sequences_checked"] == 33 and pr["outside_tolerance"] == 1 and pr["departures_elsewhere"] == 0
    assert back["single_sequence_ms_per_frame"] == 0.2801 and back["scaling_measured"] is False
    assert back["host_surface"] == {"single_camera_fps": 2612.3, "objects_8_fps": 14012.9, "objects_64_fps": 48211.0, "what": "h" * 70}
    assert len(back["kernel_us_per_step_top"]) == 5 and back["extras_file"] == "bench_extras.json"
    # a run without the optional legs (--cpu-frames 0, N > 1 without a pose check) still prints the keys, as null
    bare = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "config")}
    lb = bench.compact_line(bare)
    assert lb["roofline"] is None and lb["cpu_baseline"] is None and lb["pose_rmse"] is None and len(json.dumps(lb)) < 2048


def test_emit_writes_the_full_record_beside_the_line(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _canned_full_record()
    bench.emit(full)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < bench.LINE_LIMIT and json.loads(out[0])["value"] == full["value"]
    saved = json.load(open(tmp_path / "bench_extras.json"))
    assert saved["heterogeneous"]["value"] == 92819.1 and len(saved["roofline_kernels"]) == 16


def test_tri_v_is_tri_elementwise():
    k = np.arange(0, 200)
    for n in (2, 6, 24):
        assert [bench.tri(int(x), n) for x in k] == bench.tri_v(k, n).tolist()


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


def test_live_counter_passes_replace_the_committed_constants(tmp_path, monkeypatch):
    """bench.live_pmc_passes: the command runs itself under `rocprofv3 --kernel-trace --pmc ...` three times (FETCH_SIZE, WRITE_SIZE,
    SQ_ACTIVE_INST_VALU + GRBM_GUI_ACTIVE) and uses what those passes counted.  Here rocprofv3 is a stand-in script that writes the csv
    a pass would leave (test infrastructure: no device), so the parsing, the per-kernel means, the stamp and the fall-back are covered;
    the real thing runs on the GPU box in the default bench command."""
    fake = tmp_path / "bin"
    fake.mkdir()
    (fake / "rocprofv3").write_text('''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
d = a[a.index("-d") + 1]
counters = a[a.index("--pmc") + 1:a.index("--output-format")]
if os.environ.get("FAKE_FAIL") in counters:
    sys.exit(3)
if os.environ.get("FAKE_HANG") in counters:
    import time
    time.sleep(600)
os.makedirs(os.path.join(d, "host", "1"), exist_ok=True)
vals = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0, "SQ_ACTIVE_INST_VALU": 256.0 * 1000, "GRBM_GUI_ACTIVE": 8.0 * 2000}
with open(os.path.join(d, "host", "1", "pmc_counter_collection.csv"), "w") as f:
    f.write('"Kernel_Name","Counter_Name","Counter_Value"\\n')
    for kern, scale in (("void edgehip::k_stage_a_fused<752, 5>(edgehip::FusedArgs)", 1.0), ("edgehip::k_try_velrot(edgehip::TvrArgs)", 2.0)):
        for rep in range(3):
            for c in counters:
                f.write('"%s","%s",%r\\n' % (kern, c, vals[c] * scale))
print('noise {"metric": "frames_per_second", "config": {"keylines_per_frame_timed_mean": 14321.5}}')
''')
    os.chmod(fake / "rocprofv3", 0o755)
    monkeypatch.setenv("PATH", str(fake) + os.pathsep + os.environ["PATH"])
    monkeypatch.setattr(bench, "LIVE_PMC", {"used": False, "note": "not attempted", "issue": None})
    bench._PMC_CACHE.clear()
    try:
        assert bench.live_pmc_passes(["--steps", "2"], 1024, budget_s=120)
        assert bench.LIVE_PMC["used"] and bench.LIVE_PMC["kernels"] == 2
        assert bench.pmc_kn() == 14321.5
        assert bench.pmc_counters("A.fused", 1024) == (1000.0 * 1024, 500.0 * 1024)          # KiB -> bytes, per launch
        assert bench.pmc_counters("B.try_velrot", 1024) == (2000.0 * 1024, 1000.0 * 1024)
        assert bench.pmc_counters("A.fused", 512) is None                                     # another batch size: not these counters
        # 4 x 256000 / 1024 quad-cycles busy over 2000 cycles
        assert bench.issue_fracs()["A.fused"] == 0.5 and bench.issue_fracs()["B.try_velrot"] == 0.5
        # a failing SQ pass keeps the live HBM counters and says so; a failing FETCH pass keeps the committed constants
        bench._PMC_CACHE.clear()
        monkeypatch.setattr(bench, "LIVE_PMC", {"used": False, "note": "not attempted", "issue": None})
        monkeypatch.setenv("FAKE_FAIL", "GRBM_GUI_ACTIVE")
        assert bench.live_pmc_passes(["--steps", "2"], 1024, budget_s=120)
        assert bench.LIVE_PMC["issue"] is None and "exited 3" in bench.LIVE_PMC["issue_note"]
        bench._PMC_CACHE.clear()
        monkeypatch.setattr(bench, "LIVE_PMC", {"used": False, "note": "not attempted", "issue": None})
        monkeypatch.setenv("FAKE_FAIL", "FETCH_SIZE")
        assert not bench.live_pmc_passes(["--steps", "2"], 1024, budget_s=120)
        assert not bench.LIVE_PMC["used"] and "exited 3" in bench.LIVE_PMC["note"] and True not in bench._PMC_CACHE
        # a pass that hangs is ended with its whole process group when the budget is up, and the committed constants stay
        monkeypatch.delenv("FAKE_FAIL")
        monkeypatch.setenv("FAKE_HANG", "WRITE_SIZE")
        monkeypatch.setattr(bench, "LIVE_PMC", {"used": False, "note": "not attempted", "issue": None})
        t0 = time.time()
        assert not bench.live_pmc_passes(["--steps", "2"], 1024, budget_s=22)
        assert time.time() - t0 < 40 and "did not finish" in bench.LIVE_PMC["note"] and True not in bench._PMC_CACHE
    finally:
        bench._PMC_CACHE.clear()
