"""GPU: the C++ rebvo::REBVO mirror end to end — frames enter through requestCustomCamBuffer, results leave
through the output callback and the TUM trajectory file — against the reference oracle on the same frames.

Delivery rule mirrored from the reference's 4-player ring (rebvo_second_t.cpp:622-623): frame j reaches the
callback after frame j+1 was tracked, carrying its own nav record and its edge map as the tracker left it
(rotated by frame j+1's estimate); the last frame is never delivered."""
import os
import subprocess

import numpy as np
import pytest

from rebvo_amd import edgehip, synth
from tests.helpers import write_global_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rebvo_amd", "lib", "custom_cam_replay")


def test_custom_cam_replay_matches_reference(tmp_path):
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    if not os.path.exists(EXE):
        pytest.fail("custom_cam_replay not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    w, h, n, t0, dt = 376, 240, 9, 1.0, 0.05
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n)]
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    cfg, dump, tray = tmp_path / "cfg", tmp_path / "dump.txt", tmp_path / "tray.txt"
    write_global_config(cfg, edgehip.euroc_params(w, h), log_file=str(tmp_path / "log.m"), tray_file=str(tray), save_log=1)
    r = subprocess.run([EXE, str(cfg), str(tmp_path / "frames.rgb24"), str(n), str(t0), str(dt), str(dump)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = np.loadtxt(dump, ndmin=2)
    assert len(rows) == n - 1, r.stdout                    # the last frame is never delivered
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
    navs = []
    path = 0.0
    for k, f in enumerate(frames):
        _, nav = orc.process_frame(f, t0 + dt * k)
        navs.append(nav)
        if k == 0:
            continue
        j = k - 1                                          # frame delivered once frame k has been tracked
        row = rows[j]
        assert int(row[0]) == j and abs(row[1] - (t0 + dt * j)) < 1e-12
        kl = orc.keylines(j % 8)                           # old slot, after rotate_keylines/FordwardMatch of frame k
        assert int(row[2]) == len(kl)
        if j > 0:
            assert int(row[4]) == navs[j].estimation_ok
            path += np.linalg.norm(navs[j].V[:])
            assert np.allclose(row[5:8], navs[j].Pos[:], atol=1e-6 * path + 1e-9)
            assert np.allclose(row[8:11], navs[j].PoseLie[:], atol=1e-7)
            assert np.allclose(row[11:14], navs[j].Vel[:], rtol=1e-5, atol=1e-9)
        assert abs(row[14] - kl["rho"].sum()) <= 1e-6 * abs(kl["rho"].sum()) + 1e-9
        assert abs(row[15] - kl["s_rho"].sum()) <= 1e-6 * abs(kl["s_rho"].sum()) + 1e-9
    # TUM trajectory: one line per delivered frame: t x y z qx qy qz qw
    tr = np.loadtxt(tray, ndmin=2)
    assert tr.shape == (n - 1, 8)
    assert np.allclose(tr[:, 0], t0 + dt * np.arange(n - 1))
    assert np.allclose(np.linalg.norm(tr[:, 4:8], axis=1), 1.0)
    assert np.allclose(tr[-1, 1:4], navs[n - 2].Pos[:], atol=1e-6 * path + 1e-9)


def test_processor_config_places_the_threads(tmp_path):
    """&ProcesorConfig SetAffinity / CamaraT1 / CamaraT3 (src/rebvo/rebvo.cpp:101-104, rebvo_first_t.cpp:136-141,
    rebvo_third_t.cpp:54-59): the tracking and the output thread are pinned where the config says — same results —
    and a CPU that does not exist stops the object with the reference's message."""
    if not os.path.exists(EXE):
        pytest.fail("custom_cam_replay not built — a broken snapshot: run __graft_entry__.build()")
    w, h, n, t0, dt = 376, 240, 6, 1.0, 0.05
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n)]
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    cpus = sorted(os.sched_getaffinity(0))
    outs = []
    for tag, aff in (("free", None), ("pinned", (cpus[0], cpus[0], cpus[-1]))):
        cfg, dump = tmp_path / f"cfg_{tag}", tmp_path / f"dump_{tag}.txt"
        write_global_config(cfg, edgehip.euroc_params(w, h), affinity=aff)
        r = subprocess.run([EXE, str(cfg), str(tmp_path / "frames.rgb24"), str(n), str(t0), str(dt), str(dump)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "Cannot set cpu affinity" not in r.stdout, r.stdout + r.stderr
        outs.append(open(dump).read())
    assert outs[0] == outs[1] and len(outs[0].splitlines()) == n - 1
    cfg = tmp_path / "cfg_bad"
    write_global_config(cfg, edgehip.euroc_params(w, h), affinity=(100000, 0, 100000))
    r = subprocess.run([EXE, str(cfg), str(tmp_path / "frames.rgb24"), str(n), str(t0), str(dt), str(tmp_path / "dump_bad.txt")],
                       capture_output=True, text=True, timeout=300)
    assert "Cannot set cpu affinity" in r.stdout, r.stdout + r.stderr
    assert r.returncode != 0 or len(open(tmp_path / "dump_bad.txt").read().splitlines()) < n - 1
