"""GPU: the IMU branch end to end.  An EuRoC-layout data set with a synthetic IMU csv
(ImuMode=2) goes through the host library: DataSetCam + ImuGrabber, gyro pre-rotation, Minimizer_V / ExtRotVel and
the mapper on the GPU, BiasCorrect and the scale filter on the host.  The oracle runs the reference's own classes in
the ImuMode > 0 order of rebvo_second_t.cpp (ref_process_frame_imu) on the same frames and the same integrated IMU
data, in a fresh process (its acceleration histories are process-wide statics).

Tolerance: the GPU sums fp64 in tree order and solves the 6x6 / 7x7 systems by Jacobi instead of LAPACK -> 1e-6
relative on velocities / rotations, 1e-5 on the filter state."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from rebvo_amd import edgehip, synth
from tests.helpers import write_global_config

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rebvo_amd", "lib", "dataset_replay")

W, H, N = 376, 240, 22
DT_NS = 50_000_000
T0_NS = 1403636579763555584
IMU_NS = 5_000_000
INIT_BIAS_FRAMES = 3


def _so3(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def _write_dataset(tmp_path):
    seq = list(synth.billboard_sequence(W, H, N))
    frames = [f for f, _, _ in seq]
    tw = synth.smooth_trajectory(N, 13)
    cam0 = tmp_path / "mav0" / "cam0"
    (cam0 / "data").mkdir(parents=True)
    t_ns = [T0_NS + DT_NS * k for k in range(N)]
    with open(cam0 / "data.csv", "w") as f:
        f.write("#timestamp [ns],filename\n")
        for k, fr in enumerate(frames):
            PIL.fromarray(fr[:, :, 0], "L").save(cam0 / "data" / f"{t_ns[k]}.png")
            f.write(f"{t_ns[k]},{t_ns[k]}.png\n")
    # IMU: the camera turns by exp(tw_rot[k]) (points) between frames k and k+1, i.e. the body rate is -tw_rot/dt; a
    # constant gyro bias and a little noise on top; the accelerometer sees gravity in the camera frame.
    rng = np.random.default_rng(21)
    bias = np.array([0.004, -0.002, 0.003])
    imu_csv = tmp_path / "imu.csv"
    with open(imu_csv, "w") as f:
        f.write("#timestamp [ns],w_RS_S_x,w_y,w_z,a_x,a_y,a_z\n")
        ts = T0_NS - 12 * IMU_NS
        while ts < t_ns[-1] + 20 * IMU_NS:
            k = min(max((ts - T0_NS) // DT_NS, 0), N - 1)
            gyro = -tw[int(k), 3:] / (DT_NS * 1e-9) + bias + rng.normal(size=3) * 1e-4
            g_cam = seq[int(k)][1] @ np.array([0.0, 9.8, 0.0])
            acc = -g_cam + rng.normal(size=3) * 2e-3
            f.write("%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n" % (ts, *gyro, *acc))
            ts += IMU_NS
    se3 = tmp_path / "se3.csv"
    Rci, Tci = _so3(np.array([0.01, -0.02, 0.015])), np.array([0.02, -0.01, 0.005])
    with open(se3, "w") as f:
        for i in range(3):
            f.write(",".join("%.17g" % v for v in Rci[i]) + ",%.17g,\n" % Tci[i])
    return frames, t_ns, cam0, imu_csv, se3


def test_imu_mode2_replay_matches_reference(tmp_path):
    from oracle import oracle
    if not oracle.available("ref") or not os.path.exists(EXE):
        pytest.fail("needs oracle/_ref and dataset_replay" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    frames, t_ns, cam0, imu_csv, se3 = _write_dataset(tmp_path)
    cfg, dump = tmp_path / "cfg", tmp_path / "dump.txt"
    p = edgehip.euroc_params(W, H)
    write_global_config(cfg, p, log_file=str(tmp_path / "log.m"), tray_file=str(tmp_path / "tray.txt"), save_log=1, camera_type=2,
                        dataset=(str(cam0 / "data") + "/", str(cam0 / "data.csv"), 1e-9),
                        imu=dict(mode=2, file=str(imu_csv), se3=str(se3), time_scale=1e-9, InitBiasFrameNum=INIT_BIAS_FRAMES))
    r = subprocess.run([EXE, str(cfg), str(dump)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rows = np.loadtxt(dump, ndmin=2)
    assert len(rows) == N - 1, r.stdout[-2000:]

    # the integrated IMU data of every frame, from the REFERENCE's grabber (no process-wide state there)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libreforacle.so"), mode=C.RTLD_GLOBAL)
    ref.ref_imu_grabber_load.restype = C.c_void_p
    g = C.c_void_p(ref.ref_imu_grabber_load(str(imu_csv).encode(), C.c_double(1e-9)))
    assert g.value and ref.ref_imu_grabber_load_se3(g, str(se3).encode()) == 1
    t = [float(np.float64(v) * 1e-9) for v in t_ns]
    imu_rows, t_prev = [], 0.0
    for k in range(N):
        d = oracle.ImuIntegrated()
        ref.ref_imu_grabber_grab(g, C.c_double(t_prev), C.c_double(t[k]), C.byref(d))
        assert d.n > 0
        imu_rows.append(d.as_row())
        t_prev = t[k]
    o = _run_reference(tmp_path, frames, t, imu_rows, {"init_bias_frame_num": INIT_BIAS_FRAMES})
    _compare(rows, o)
    _compare_log(tmp_path / "log.m", len(rows), imu_rows, o)


# every key the reference's ThirdThread writes per frame (rebvo_third_t.cpp:264-304), in its order
LOG_KEYS = ["Kp", "RKp", "Rot", "Vel", "RotGiro", "t", "dt", "i", "Pose", "Pos", "K", "KLN", "Giro", "Acel", "CAcel", "DGiro", "GBias",
            "dWv", "dWgv", "g", "VBias", "Av", "As", "Posgv", "SMM", "TProc0", "TProc1", "TProc2"]


def _parse_log(path):
    import re
    out, order = {}, []
    pat = re.compile(r"^(\w+)_cv\((\d+),[:,]+\)=\[?([^\]]*)\]?;$")
    for line in open(path):
        m = pat.match(line.strip())
        assert m, line
        key, idx = m.group(1), int(m.group(2))
        vals = [float(v) for v in re.split(r"[;,]", m.group(3))]
        out.setdefault(key, {})[idx] = np.array(vals)
        if idx == 1:
            order.append(key)
    return out, order


def _compare_log(path, nrows, imu_rows, o):
    """The .m log of the same run: all 28 keys of the reference's writer, in its order, one entry per delivered frame; the IMU
    block (the part round 2 did not write) against the reference's own values, at the log's 6 significant digits."""
    log, order = _parse_log(path)
    assert order == LOG_KEYS, order
    for key in LOG_KEYS:
        assert sorted(log[key]) == list(range(1, nrows + 1)), key
    close = lambda a, b: np.allclose(a, b, rtol=2e-5, atol=2e-7)
    for k in range(1, nrows - 1):                 # log entry k+1 = frame k
        e = {key: log[key][k + 1] for key in LOG_KEYS}
        r = np.asarray(imu_rows[k])
        assert close(e["Giro"], r[11:14]) and close(e["Acel"], r[14:17]) and close(e["DGiro"], r[20:23]) and close(e["CAcel"], r[23:26]), k
        assert close(e["GBias"], o["Bg"][k]) and close(e["dWv"], o["dWv"][k]) and close(e["RotGiro"], o["RotGiro"][k]), k
        assert close(e["g"], o["g"][k]) and close(e["VBias"], o["b_est"][k]), k
        assert close(e["Av"], o["Av"][k]) and close(e["As"], o["As"][k]), k
        assert int(e["SMM"][0]) == 0 and int(e["KLN"][0]) > 0


def _compare(rows, o):
    def close(a, b, rtol, atol):
        return np.allclose(a, b, rtol=rtol, atol=atol)

    filter_frames = 0
    for k in range(1, N - 1):                     # row k = frame k, delivered after frame k+1 was tracked
        row = rows[k]
        assert int(row[0]) == k
        assert int(row[2]) == int(o["klprev_n"][k + 1])                      # KeyLine count of frame k's edge map
        assert int(row[4]) == int(o["estimation_ok"][k]) and int(row[3]) == int(o["klm_num"][k]), k
        assert abs(row[48] - o["dt"][k]) < 1e-12
        # Vel = Vg * scale: it inherits the scale filter's tolerance (the filter's 7x7 / 11-row Gauss-Newton solves are
        # ill-conditioned; LAPACK's SVD in the reference, Jacobi here: over 60 frames the scale drifts apart by up to
        # 1.3e-6 and comes back, while every per-KeyLine quantity — Vg, Bg, RotLie, the depth sums, the match counts —
        # stays within 1e-13: tools/experiments/exp_imu_long.py)
        assert close(row[11:14], o["Vel"][k], 1e-5, 1e-7), (k, row[11:14], o["Vel"][k])
        assert close(row[16:19], o["RotLie"][k], 1e-6, 1e-8), (k, row[16:19], o["RotLie"][k])
        assert close(row[19:22], o["RotGiro"][k], 1e-6, 1e-7)
        assert close(row[29:32], o["Vg"][k], 1e-6, 1e-9) and close(row[32:35], o["Bg"][k], 1e-6, 1e-10)
        assert close(row[5:8], o["Pos"][k], 1e-5, 1e-7) and close(row[8:11], o["PoseLie"][k], 1e-5, 1e-7)
        assert close(row[25:29], [o["scale"][k], o["K"][k], o["Kp"][k], o["RKp"][k]], 1e-5, 1e-12)
        assert close(row[35:42], o["X"][k], 1e-5, 1e-8) and close(row[22:25], o["g"][k], 1e-5, 1e-7)
        assert close(row[45:48], o["u_est"][k], 1e-5, 1e-8)
        sr = o["klprev_rho_sum"][k + 1]
        assert abs(row[14] - sr) <= 1e-6 * abs(sr) + 1e-9
        filter_frames += int(o["scale"][k] != 1.0)
    assert int(o["estimation_ok"][1:N - 1].sum()) >= N - 6     # the sequence actually tracks
    assert int(o["init"][N - 2]) == 1 and filter_frames >= 5   # bias start-up finished, scale filter ran
    assert np.abs(o["Pos"][N - 2]).max() > 0                   # and the IMU pose integration moved


def _run_reference(tmp_path, frames, t, imu_rows, imu_over):
    inp, outp = tmp_path / "in.npz", tmp_path / "out.npz"
    np.savez(inp, frames=np.stack(frames), t=np.array(t), imu=np.stack(imu_rows), w=W, h=H, over=json.dumps({}),
             imu_over=json.dumps(imu_over))
    rr = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "imu_ref_runner.py"), "sequence", str(inp), str(outp)],
                        capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert rr.returncode == 0, rr.stdout[-2000:] + rr.stderr[-2000:]
    return np.load(outp)


def test_imu_mode1_pushimu_matches_reference(tmp_path):
    """ImuMode=1: the application pushes samples (REBVO::pushIMU) while it feeds frames through the custom-camera ring;
    a 96-slot sample ring wraps several times over the run."""
    from oracle import oracle
    exe = os.path.join(ROOT, "rebvo_amd", "lib", "custom_cam_replay")
    if not oracle.available("ref") or not os.path.exists(exe):
        pytest.fail("needs oracle/_ref and custom_cam_replay" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    frames, t_ns, cam0, imu_csv_ns, se3 = _write_dataset(tmp_path)
    t0, dt = 10.0, 0.05
    t = [t0 + dt * k for k in range(N)]
    # the same samples, re-stamped in seconds relative to the custom camera's clock
    raw = np.loadtxt(imu_csv_ns, delimiter=",", comments="#")
    raw[:, 0] = t0 + (raw[:, 0] - T0_NS) * 1e-9
    imu_csv = tmp_path / "imu_s.csv"
    with open(imu_csv, "w") as f:
        for r in raw:
            f.write(",".join("%.17g" % v for v in r) + "\n")
    raw = np.loadtxt(imu_csv, delimiter=",")            # what both sides parse
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    cfg, dump = tmp_path / "cfg", tmp_path / "dump.txt"
    write_global_config(cfg, edgehip.euroc_params(W, H), camera_type=3,
                        imu=dict(mode=1, se3=str(se3), InitBiasFrameNum=INIT_BIAS_FRAMES, SampleTime=0.005, CircBufferSize=96))
    r = subprocess.run([exe, str(cfg), str(tmp_path / "frames.rgb24"), str(N), repr(t0), repr(dt), str(dump), str(imu_csv)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rows = np.loadtxt(dump, ndmin=2)
    assert len(rows) == N - 1, r.stdout[-2000:]
    # reference grabber fed the same way (a big ring: what is compared is the integration, not the ring size)
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libreforacle.so"), mode=C.RTLD_GLOBAL)
    ref.ref_imu_grabber_new.restype = C.c_void_p
    g = C.c_void_p(ref.ref_imu_grabber_new(4096, C.c_double(0.005)))
    assert ref.ref_imu_grabber_load_se3(g, str(se3).encode()) == 1
    for rw in raw:
        gy, ac = np.ascontiguousarray(rw[1:4]), np.ascontiguousarray(rw[4:7])
        assert ref.ref_imu_grabber_push(g, C.c_double(rw[0]), gy.ctypes.data_as(C.POINTER(C.c_double)),
                                        ac.ctypes.data_as(C.POINTER(C.c_double))) == 1
    imu_rows, t_prev = [], 0.0
    for k in range(N):
        d = oracle.ImuIntegrated()
        ref.ref_imu_grabber_grab(g, C.c_double(t_prev), C.c_double(t[k]), C.byref(d))
        assert d.n > 0
        imu_rows.append(d.as_row())
        t_prev = t[k]
    o = _run_reference(tmp_path, frames, t, imu_rows, {"init_bias_frame_num": INIT_BIAS_FRAMES})
    _compare(rows, o)


def test_imu_branch_batched_on_the_device_matches_reference(tmp_path):
    _batched_branch(tmp_path, 8, range(8))


@pytest.mark.parametrize("w,h,B,check", [(752, 480, 8, (0, 3, 7)), (376, 240, 200, (0, 5, 67, 199)), (752, 480, 192, (1, 190))])
def test_imu_branch_batched_at_the_baseline_size_and_behind_the_one_kernel_stage_a(tmp_path, monkeypatch, w, h, B, check):
    """The same replay at BASELINE's 752x480, and with >= 192 sequences per launch, where stage A is the one-kernel form
    (k_stage_a_fused: EDGEHIP_FUSED_MIN_BATCH) in front of the IMU branch; a few sequences of the batch against the reference."""
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "W", w)
    monkeypatch.setattr(mod, "H", h)
    _batched_branch(tmp_path, B, check)


def _batched_branch(tmp_path, B, check):
    """ImuMode > 0 for a whole batch inside edgehip_process_frame (edgehip_imu_enable / edgehip_set_imu): eight sequences —
    the same data set entered 0..7 frames late, so that bias start-up, scale filter and map are in a different state in
    every one of them at any time — advance in lock-step with no host synchronisation between the stages; the filters run
    on the device, one thread per sequence.  Every sequence against the reference's own ImuMode > 0 frame order
    (ref_process_frame_imu, a fresh process per sequence: its acceleration histories are process-wide statics)."""
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("needs oracle/_ref" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    global N
    n_run = N
    n_all = n_run + 8 - 1      # sequence s enters the data set (s % 8) frames late
    N_keep = N
    try:
        N = n_all
        frames, t_ns, cam0, imu_csv, se3 = _write_dataset(tmp_path)
    finally:
        N = N_keep
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libreforacle.so"), mode=C.RTLD_GLOBAL)
    ref.ref_imu_grabber_load.restype = C.c_void_p
    g = C.c_void_p(ref.ref_imu_grabber_load(str(imu_csv).encode(), C.c_double(1e-9)))
    assert g.value and ref.ref_imu_grabber_load_se3(g, str(se3).encode()) == 1
    t = [float(np.float64(v) * 1e-9) for v in t_ns]
    imu_rows, t_prev = [], 0.0
    for k in range(n_all):
        d = oracle.ImuIntegrated()
        ref.ref_imu_grabber_grab(g, C.c_double(t_prev), C.c_double(t[k]), C.byref(d))
        assert d.n > 0
        imu_rows.append(d.as_row())
        t_prev = t[k]

    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=B, nslots=3)
    eh.imu_enable(edgehip.euroc_imu_params(init_bias_frame_num=INIT_BIAS_FRAMES))
    got = []
    for k in range(n_run):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[k + s % 8] for s in range(B)]))
        eh.set_imu([edgehip.ImuIntegrated.from_row(imu_rows[k + s % 8]) for s in range(B)])
        eh.process_frame(np.array([t[k + s % 8] for s in range(B)]))
        got.append((eh.read_nav(), eh.read_nav_imu()))
    eh.close()

    def close(a, b, rtol, atol):
        return np.allclose(np.array(a[:]) if hasattr(a, "__len__") else a, b, rtol=rtol, atol=atol)

    for s in check:
        sub = tmp_path / f"seq{s}"
        sub.mkdir()
        s8 = s % 8
        o = _run_reference(sub, frames[s8:s8 + n_run], t[s8:s8 + n_run], imu_rows[s8:s8 + n_run], {"init_bias_frame_num": INIT_BIAS_FRAMES})
        filter_frames = 0
        for k in range(1, n_run):
            nav, ni = got[k][0][s], got[k][1][s]
            assert ni.kn == int(o["kn"][k]) and nav.kn == ni.kn, (s, k)
            assert ni.estimation_ok == int(o["estimation_ok"][k]) and ni.klm_num == int(o["klm_num"][k]), (s, k, ni.klm_num, o["klm_num"][k])
            assert ni.init == int(o["init"][k]) and abs(ni.dt - o["dt"][k]) < 1e-12
            assert close(ni.Vg, o["Vg"][k], 1e-6, 1e-9) and close(ni.Bg, o["Bg"][k], 1e-6, 1e-10), (s, k, ni.Vg[:], o["Vg"][k])
            assert close(ni.dVv, o["dVv"][k], 1e-5, 1e-8) and close(ni.dWv, o["dWv"][k], 1e-5, 1e-9), (s, k)
            assert close(ni.Vgv, o["Vgv"][k], 1e-6, 1e-9) and close(ni.RotLie, o["RotLie"][k], 1e-6, 1e-8), (s, k)
            assert close(ni.RotGiro, o["RotGiro"][k], 1e-6, 1e-7) and close(ni.Vel, o["Vel"][k], 1e-5, 1e-7), (s, k)
            assert close(ni.Av, o["Av"][k], 1e-6, 1e-8) and close(ni.As, o["As"][k], 1e-9, 1e-12), (s, k)
            assert close([ni.scale, ni.K, ni.Kp, ni.RKp], [o["scale"][k], o["K"][k], o["Kp"][k], o["RKp"][k]], 1e-5, 1e-12), (s, k)
            # filter state: scale angle, gravity (|g| = 9.8: components near zero are compared against the vector's size, the
            # 7x7 / 11-row Gauss-Newton solves are ill-conditioned and solved by Jacobi here, LAPACK there), visual bias
            Xg, Xr = np.array(ni.X[:]), np.array(o["X"][k])
            assert abs(Xg[0] - Xr[0]) <= 1e-5 * abs(Xr[0]) + 1e-9, (s, k, Xg, Xr)
            assert np.allclose(Xg[1:4], Xr[1:4], rtol=0, atol=1e-7 * 9.8), (s, k, Xg, Xr)
            assert np.allclose(Xg[4:7], Xr[4:7], rtol=1e-4, atol=1e-8), (s, k, Xg, Xr)
            assert np.allclose(ni.g[:], o["g"][k], rtol=0, atol=1e-7 * 9.8) and close(ni.u_est, o["u_est"][k], 1e-5, 1e-7), (s, k)
            assert close(ni.Pos, o["Pos"][k], 1e-5, 1e-7) and close(ni.PoseLie, o["PoseLie"][k], 1e-5, 1e-7), (s, k)
            assert close(ni.Vgva, o["Vgva"][k], 1e-5, 1e-8) and close(ni.b_est, o["b_est"][k], 1e-4, 1e-9)
            assert abs(ni.s_rho_q - o["s_rho_q"][k]) < 1e-9
            filter_frames += int(o["scale"][k] != 1.0)
        assert int(o["estimation_ok"][1:n_run].sum()) >= n_run - 6 and filter_frames >= 4, (s, filter_frames)


def test_eight_imu_objects_in_one_batch_group(tmp_path):
    """ImuMode = 2 behind the plugin surface as ONE batch (round 6: `&GPU BatchGroup` admits ImuMode 1 / 2 members): eight rebvo::REBVO
    objects, custom cameras, one IMU file; object i enters the common time line i frames late (surface_replay --stagger), so bias
    start-up, scale filter and map are in a different state in every member at any step.  The group's thread grabs every member's
    inter-frame IMU data (ImuGrabber::GrabAndIntegrate, src/UtilLib/imugrabber.cpp:178-268), hands the batch to edgehip_set_imu and the
    device runs SecondThread's ImuMode > 0 branch for all of them (rebvo_second_t.cpp:182-336, 519-544).  Checked: every object's callback
    rows (a) bit-identical to the same eight sequences run as one ctypes IMU batch fed the same integrated records, and (b) against the
    reference's own ImuMode > 0 frame order within the bounds of the tests above."""
    from oracle import oracle
    exe = os.path.join(ROOT, "rebvo_amd", "lib", "surface_replay")
    if not oracle.available("ref") or not os.path.exists(exe):
        pytest.fail("needs oracle/_ref and surface_replay — a broken snapshot: run __graft_entry__.build()")
    global N
    B, n_run, N_keep = 8, N, N
    n_all = n_run + B - 1
    try:
        N = n_all
        frames, t_ns, cam0, imu_csv_ns, se3 = _write_dataset(tmp_path)
    finally:
        N = N_keep
    t0, dt = 10.0, 0.05
    t = [t0 + dt * k for k in range(n_all)]                      # (the expression surface_replay --stagger evaluates)
    raw = np.loadtxt(imu_csv_ns, delimiter=",", comments="#")
    raw[:, 0] = t0 + (raw[:, 0] - T0_NS) * 1e-9
    imu_csv = tmp_path / "imu_s.csv"
    with open(imu_csv, "w") as f:
        for r in raw:
            f.write(",".join("%.17g" % v for v in r) + "\n")
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    cfg = tmp_path / "cfg"
    write_global_config(cfg, edgehip.euroc_params(W, H), camera_type=3, gpu=dict(group="imu8", size=B),
                        imu=dict(mode=2, file=str(imu_csv), se3=str(se3), time_scale=1.0, InitBiasFrameNum=INIT_BIAS_FRAMES))
    prefix = tmp_path / "run"
    r = subprocess.run([exe, str(cfg), str(tmp_path / "frames.rgb24"), str(n_all), str(B), str(n_run), repr(t0), repr(dt), "--stagger",
                        "--dump", str(prefix), "--threads", "3"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, REBVO_GROUP_TIMING="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    js = json.loads(r.stdout.strip().splitlines()[-1])
    assert js["objects"] == B and js["callbacks"] == B * (n_run - 1), r.stdout[-1500:]
    assert f"group 'imu8': {n_run} steps" in r.stdout, r.stdout[-1500:]          # ONE context, one launch set per step for the eight
    dumps = [np.loadtxt(f"{prefix}.{i}.txt", ndmin=2) for i in range(B)]

    # (a) the same eight sequences as one ctypes IMU batch, fed by the host library's own grabber (what the group's thread calls)
    host = C.CDLL(os.path.join(ROOT, "rebvo_amd", "lib", "librebvohost.so"))
    host.rebvo_imu_grabber_load.restype = C.c_void_p
    ghs = [C.c_void_p(host.rebvo_imu_grabber_load(str(imu_csv).encode(), C.c_double(1.0))) for _ in range(B)]   # a grabber reads forward only: one per sequence
    assert all(g_.value and host.rebvo_imu_grabber_load_se3(g_, str(se3).encode()) == 1 for g_ in ghs)

    def grab(s_, ta, tb):
        d = edgehip.ImuIntegrated()
        host.rebvo_imu_grabber_grab(ghs[s_], C.c_double(ta), C.c_double(tb), C.byref(d))
        assert d.n > 0
        return d
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=B, nslots=3)
    eh.imu_enable(edgehip.euroc_imu_params(init_bias_frame_num=INIT_BIAS_FRAMES))      # the &IMU section write_global_config wrote
    got, kls = [], []
    for k in range(n_run):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[k + s] for s in range(B)]))
        eh.set_imu([grab(s, t[k + s - 1] if k else 0.0, t[k + s]) for s in range(B)])
        eh.process_frame(np.array([t[k + s] for s in range(B)]))
        got.append((eh.read_nav(), eh.read_nav_imu()))
        if k:
            kls.append([eh.download_keylines(s, (eh.cur_slot() + 2) % 3, want_mask=False)[0] for s in range(B)])
    eh.close()
    arr = lambda v: np.array(v[:])
    for s in range(B):
        rows = dumps[s]
        assert len(rows) == n_run - 1
        for j in range(n_run - 1):                 # frame j is delivered once frame j + 1 has been tracked
            row, kl = rows[j], kls[j][s]
            assert int(row[0]) == j and row[1] == t[j + s] and int(row[2]) == len(kl)
            assert row[14] == np.cumsum(kl["rho"])[-1] and row[15] == np.cumsum(kl["s_rho"])[-1], (s, j)
            if j == 0:
                continue
            nav, ni = got[j][0][s], got[j][1][s]
            assert int(row[3]) == ni.klm_num and int(row[4]) == ni.estimation_ok, (s, j)
            for a, b in ((row[5:8], ni.Pos), (row[8:11], ni.PoseLie), (row[11:14], ni.Vel), (row[16:19], ni.RotLie), (row[19:22], ni.RotGiro),
                         (row[22:25], ni.g), (row[29:32], ni.Vg), (row[32:35], ni.Bg), (row[35:42], ni.X), (row[42:45], ni.b_est), (row[45:48], ni.u_est)):
                assert np.array_equal(a, arr(b)), (s, j, a, arr(b))
            assert np.array_equal(row[25:29], [ni.scale, ni.K, ni.Kp, ni.RKp]) and row[48] == ni.dt, (s, j)

    # (b) members against the reference's ImuMode > 0 branch on their own frames, stamps and (reference-grabbed) IMU data
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libreforacle.so"), mode=C.RTLD_GLOBAL)
    ref.ref_imu_grabber_load.restype = C.c_void_p
    for s in (0, 3, 7):
        g = C.c_void_p(ref.ref_imu_grabber_load(str(imu_csv).encode(), C.c_double(1.0)))
        assert g.value and ref.ref_imu_grabber_load_se3(g, str(se3).encode()) == 1
        imu_rows, t_prev = [], 0.0
        for k in range(n_run):
            d = oracle.ImuIntegrated()
            ref.ref_imu_grabber_grab(g, C.c_double(t_prev), C.c_double(t[k + s]), C.byref(d))
            assert d.n > 0
            imu_rows.append(d.as_row())
            t_prev = t[k + s]
        sub = tmp_path / f"ref{s}"
        sub.mkdir()
        o = _run_reference(sub, frames[s:s + n_run], t[s:s + n_run], imu_rows, {"init_bias_frame_num": INIT_BIAS_FRAMES})
        # the bounds of _batched_branch (the device-side filters: gravity components against the vector's size, the visual bias at 1e-4)
        close = lambda a, b, rtol, atol: np.allclose(a, b, rtol=rtol, atol=atol)
        filter_frames = 0
        for k in range(1, n_run - 1):
            row = dumps[s][k]
            assert int(row[0]) == k and int(row[2]) == int(o["klprev_n"][k + 1])
            assert int(row[4]) == int(o["estimation_ok"][k]) and int(row[3]) == int(o["klm_num"][k]), (s, k)
            assert abs(row[48] - o["dt"][k]) < 1e-12
            assert close(row[29:32], o["Vg"][k], 1e-6, 1e-9) and close(row[32:35], o["Bg"][k], 1e-6, 1e-10), (s, k)
            assert close(row[16:19], o["RotLie"][k], 1e-6, 1e-8) and close(row[19:22], o["RotGiro"][k], 1e-6, 1e-7), (s, k)
            assert close(row[11:14], o["Vel"][k], 1e-5, 1e-7), (s, k)
            assert close(row[25:29], [o["scale"][k], o["K"][k], o["Kp"][k], o["RKp"][k]], 1e-5, 1e-12), (s, k)
            Xg, Xr = row[35:42], np.array(o["X"][k])
            assert abs(Xg[0] - Xr[0]) <= 1e-5 * abs(Xr[0]) + 1e-9, (s, k, Xg, Xr)
            assert np.allclose(Xg[1:4], Xr[1:4], rtol=0, atol=1e-7 * 9.8) and np.allclose(Xg[4:7], Xr[4:7], rtol=1e-4, atol=1e-8), (s, k, Xg, Xr)
            assert np.allclose(row[22:25], o["g"][k], rtol=0, atol=1e-7 * 9.8) and close(row[45:48], o["u_est"][k], 1e-5, 1e-7), (s, k)
            assert close(row[5:8], o["Pos"][k], 1e-5, 1e-7) and close(row[8:11], o["PoseLie"][k], 1e-5, 1e-7), (s, k)
            assert close(row[42:45], o["b_est"][k], 1e-4, 1e-9), (s, k)
            sr = o["klprev_rho_sum"][k + 1]
            assert abs(row[14] - sr) <= 1e-6 * abs(sr) + 1e-9
            filter_frames += int(o["scale"][k] != 1.0)
        assert int(o["estimation_ok"][1:n_run - 1].sum()) >= n_run - 6 and filter_frames >= 4, (s, filter_frames)
