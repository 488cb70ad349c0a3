"""GPU: CameraType=2 end to end — an EuRoC-layout dataset (mav0/cam0/data.csv with nanosecond stamps + grey PNGs)
is read by the library's own DataSetCam, tracked on the GPU and compared with the reference oracle fed the same
frames.  SURVEY.md section 8f1."""
import os
import subprocess

import numpy as np
import pytest

from rebvo_amd import edgehip, synth
from tests.helpers import write_global_config

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rebvo_amd", "lib", "dataset_replay")


def test_euroc_layout_replay(tmp_path):
    from oracle import oracle
    if not oracle.available("ref") or not os.path.exists(EXE):
        pytest.fail("needs oracle/_ref and dataset_replay" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    w, h, n = 376, 240, 8
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n)]
    cam0 = tmp_path / "mav0" / "cam0"
    (cam0 / "data").mkdir(parents=True)
    t_ns = [1403636579763555584 + 50_000_000 * k for k in range(n)]
    with open(cam0 / "data.csv", "w") as f:
        f.write("#timestamp [ns],filename\r\n")
        for k, fr in enumerate(frames):
            PIL.fromarray(fr[:, :, 0], "L").save(cam0 / "data" / f"{t_ns[k]}.png")
            f.write(f"{t_ns[k]},{t_ns[k]}.png\r\n")                  # CR LF line ends as in the real files
    cfg, dump, tray = tmp_path / "cfg", tmp_path / "dump.txt", tmp_path / "tray.txt"
    p = edgehip.euroc_params(w, h)
    write_global_config(cfg, p, log_file=str(tmp_path / "log.m"), tray_file=str(tray), save_log=1, camera_type=2,
                        dataset=(str(cam0 / "data") + "/", str(cam0 / "data.csv"), 1e-9))
    r = subprocess.run([EXE, str(cfg), str(dump)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"Loaded {n} File names" in r.stdout
    rows = np.loadtxt(dump, ndmin=2)
    assert len(rows) == n - 1
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
    path = 0.0
    for k, fr in enumerate(frames):
        t = float(np.float64(t_ns[k]) * 1e-9)                         # std::stod(line) * time_scale
        _, nav = orc.process_frame(fr, t)
        if k >= 1 and k < n:
            pass
        if k == 0:
            prev = nav
            continue
        row = rows[k - 1]                                             # frame k-1 is delivered after frame k was tracked
        assert int(row[0]) == k - 1
        kl = orc.keylines((k - 1) % 8)
        assert int(row[2]) == len(kl)
        if k - 1 > 0:
            path += np.linalg.norm(prev.V[:])
            assert np.allclose(row[5:8], prev.Pos[:], atol=1e-6 * path + 1e-9)
        assert abs(row[14] - kl["rho"].sum()) <= 1e-6 * abs(kl["rho"].sum()) + 1e-9
        prev = nav


def _write_euroc_set(root, frames, t0_ns):
    """mav0/cam0 layout (data.csv with nanosecond stamps + grey PNGs) under `root`; returns (image dir, list file, stamps)."""
    cam0 = root / "mav0" / "cam0"
    (cam0 / "data").mkdir(parents=True)
    t_ns = [t0_ns + 50_000_000 * k for k in range(len(frames))]
    with open(cam0 / "data.csv", "w") as f:
        f.write("#timestamp [ns],filename\r\n")
        for k, fr in enumerate(frames):
            PIL.fromarray(fr[:, :, 0], "L").save(cam0 / "data" / f"{t_ns[k]}.png")
            f.write(f"{t_ns[k]},{t_ns[k]}.png\r\n")
    return str(cam0 / "data") + "/", str(cam0 / "data.csv"), t_ns


def test_multi_device_replay_two_sequences(tmp_path):
    """BASELINE configs[4] on the host side in C++ (rebvo_amd/host/examples/multi_device_replay.cpp): N data sets at once, one
    rebvo::REBVO per sequence, sequence i on device i % devices.  Two different sequences (both land on the one device a test
    box has) must each follow the reference fed the same files — i.e. the two objects, their threads and their contexts do
    not disturb each other."""
    from oracle import oracle
    exe = os.path.join(ROOT, "rebvo_amd", "lib", "multi_device_replay")
    if not oracle.available("ref") or not os.path.exists(exe):
        pytest.fail("needs oracle/_ref and multi_device_replay" " — a broken snapshot, not a reason to skip: run __graft_entry__.build()")
    w, h, n = 376, 240, 7
    p = edgehip.euroc_params(w, h)
    cfgs, sets = [], []
    for i in range(2):
        frames = [f for f, _, _ in synth.billboard_sequence(w, h, n, seed=11 + 5 * i)]
        d, lst, t_ns = _write_euroc_set(tmp_path / f"seq{i}", frames, 1403636579763555584 + 7 * i)
        cfg = tmp_path / f"cfg{i}"
        write_global_config(cfg, p, log_file=str(tmp_path / f"log{i}.m"), tray_file=str(tmp_path / f"tray{i}.txt"), save_log=0,
                            camera_type=2, dataset=(d, lst, 1e-9))
        cfgs.append(str(cfg))
        sets.append((frames, t_ns))
    r = subprocess.run([exe, "--dump", str(tmp_path / "dump"), *cfgs], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "node aggregate: 2 sequences" in r.stdout
    for i, (frames, t_ns) in enumerate(sets):
        assert f"sequence {i} device 0: frames delivered {n - 1}" in r.stdout, r.stdout
        rows = np.loadtxt(str(tmp_path / "dump") + f"{i}.txt", ndmin=2)
        assert len(rows) == n - 1
        orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
        path, prev = 0.0, None
        for k, fr in enumerate(frames):
            _, nav = orc.process_frame(fr, float(np.float64(t_ns[k]) * 1e-9))
            if k > 0:
                row = rows[k - 1]
                assert int(row[0]) == k - 1 and int(row[2]) == len(orc.keylines((k - 1) % 8))
                if k - 1 > 0:
                    path += np.linalg.norm(prev.V[:])
                    assert np.allclose(row[5:8], prev.Pos[:], atol=1e-6 * path + 1e-9)
            prev = nav
        orc.close()


def test_bench_replays_a_mounted_dataset(tmp_path):
    """bench.py --dataset: a data set in the EuRoC layout is read through the library's own DataSetCam, replayed
    as the sequences of the batch at staggered start frames inside the same timed region, and the pose check runs the CPU
    reference on the same files — `data: "euroc"` in the line.  A path that is not a data set leaves the synthetic scenes."""
    import json
    import sys
    w, h, n = 752, 480, 12
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n, seed=23)]
    _write_euroc_set(tmp_path / "MH_xx", frames, 1403636579763555584)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--nseq", "6", "--steps", "6", "--warmup", "4", "--no-extras", "--cpu-frames", "8",
           "--cpu-procs", "0"]
    env = dict(os.environ, BENCH_FORCE_MOVER="0", BENCH_EXTRAS_FILE=str(tmp_path / "bench_extras.json"))
    r = subprocess.run(cmd + ["--dataset", str(tmp_path / "MH_xx")], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, lines                      # one JSON line on stdout, of a size the driver reads
    js = json.loads(lines[0])
    assert js["data"] == "euroc" and js["config"]["dataset"]["frames_in_pool"] == n
    assert abs(js["config"]["dataset"]["mean_frame_interval_s"] - 0.05) < 1e-6
    assert js["config"]["estimation_ok"] == "6/6" and js["config"]["keylines_per_frame"] > 5000
    assert js["pose_rmse"]["sequences_checked"] == 6 and js["pose_rmse"]["departures_elsewhere"] == 0
    assert js["pose_rmse"]["position"] < 1e-6 or js["pose_rmse"]["departures_on_knife_edge_frames"] > 0
    assert js["cpu_baseline"]["kind"] == "reference" and js["cpu_baseline"]["value"] > 0
    full = json.load(open(tmp_path / "bench_extras.json"))                      # the full record beside it
    par = full["pose_rmse"]["free_running_parity"]
    assert par["sequences_checked"] == 6 and par["departures_elsewhere"] == 0 and full["config"]["dataset"]["reader"].startswith("rebvo::DataSetCam")
    r = subprocess.run(cmd + ["--dataset", str(tmp_path / "not_there"), "--cpu-frames", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    js = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert js["data"] == "synthetic" and js["config"]["dataset"] is None


def test_data_sets_as_one_batch_group_on_a_device(tmp_path):
    """multi_device_replay --group (round 5): the sequences dealt to one device form one batch group (&GPU BatchGroup): DataSetCam
    objects whose images reach the shared N-sequence context through their own camera rings, one launch set per step for all of them.
    Three data sets of different lengths (a member leaves the group when its list ends, the others carry on): every sequence's dump
    must equal, number for number, the dump of the same sequence replayed by an object of its own."""
    exe = os.path.join(ROOT, "rebvo_amd", "lib", "multi_device_replay")
    if not os.path.exists(exe):
        pytest.fail("needs multi_device_replay — a broken snapshot, not a reason to skip: run __graft_entry__.build()")
    w, h = 376, 240
    p = edgehip.euroc_params(w, h)
    cfgs, lens = [], (7, 5, 6)
    for i, n in enumerate(lens):
        frames = [f for f, _, _ in synth.billboard_sequence(w, h, n, seed=31 + 3 * i)]
        d, lst, t_ns = _write_euroc_set(tmp_path / f"seq{i}", frames, 1403636579763555584 + 11 * i)
        cfg = tmp_path / f"cfg{i}"
        write_global_config(cfg, p, camera_type=2, dataset=(d, lst, 1e-9))
        cfgs.append(str(cfg))
    outs = {}
    for tag, extra in (("alone", []), ("group", ["--group"])):
        r = subprocess.run([exe, "--devices", "1", "--dump", str(tmp_path / tag), *extra, *cfgs], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
        outs[tag] = r.stdout
        for i, n in enumerate(lens):
            assert f"sequence {i} device 0: frames delivered {n - 1}" in r.stdout, r.stdout
    for i, n in enumerate(lens):
        a = np.loadtxt(str(tmp_path / "alone") + f"{i}.txt", ndmin=2)
        g = np.loadtxt(str(tmp_path / "group") + f"{i}.txt", ndmin=2)
        assert a.shape == g.shape == (n - 1, 14) and np.array_equal(a, g), i
