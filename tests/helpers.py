"""Shared helpers of the GPU parity tests."""
import numpy as np


def to_edgehip_kl(kl):
    """oracle KEYLINE_DTYPE and edgehip KEYLINE_DTYPE are the same 168-byte layout."""
    from rebvo_amd import edgehip
    return np.frombuffer(np.ascontiguousarray(kl).tobytes(), dtype=edgehip.KEYLINE_DTYPE).copy()


def oracle_pair(w, h, n_warm, seq="billboard", **over):
    """Run the reference oracle for n_warm full frames, then stage A of the next frame.

    Returns (orc, slot_old, slot_new, nav_of_last_full_frame, frames)."""
    from oracle import oracle
    from rebvo_amd import synth
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h, **over))
    if seq == "billboard":
        frames = [f for f, _, _ in synth.billboard_sequence(w, h, n_warm + 1)]
    else:
        frames = list(synth.rects_sequence(w, h, n_warm + 1))
    nav = None
    for k in range(n_warm):
        _, nav = orc.process_frame(frames[k], 0.05 * k)
    slot_old = (n_warm - 1) % 8
    slot_new = n_warm % 8
    orc.stage_a(slot_new, frames[n_warm], nav.tresh, nav.kn)
    return orc, slot_old, slot_new, nav, frames


def inject_pair(eh, orc, slot_old, slot_new, seq=0, gslot_old=0, gslot_new=1):
    eh.upload_keylines(seq, gslot_old, to_edgehip_kl(orc.keylines(slot_old)), orc.mask(slot_old), orc.retuned(slot_old))
    eh.upload_keylines(seq, gslot_new, to_edgehip_kl(orc.keylines(slot_new)), orc.mask(slot_new), orc.retuned(slot_new))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def write_global_config(path, p, log_file="", tray_file="", save_log=0, camera_type=3, drop=(), dataset=None, imu=None, stereo=None):
    """A GlobalConfig file in the reference's format (app/rebvorun/GlobalConfig_EuRoC) from a Params struct.
    `drop` lists "Section/Key" entries to leave out (missing-key error tests).  `imu` = dict(mode=1|2, file=..., se3=...,
    time_scale=..., plus any key of the &IMU section to override) switches the IMU branch on.  `stereo` = dict(dir=..., file=...,
    ppx=, ppy=, zfx=, zfy=) sets StereoAvaiable with the pair camera's list and the &Stereo intrinsics."""
    sec = {
        "Detector": [("Sigma0", p.sigma0), ("KSigma", p.ksigma), ("ReferencePoints", p.reference_points),
                     ("MaxPoints", p.max_points), ("TrackPoints", p.track_points), ("DetectorThresh", p.detector_thresh),
                     ("DetectorAutoGain", p.auto_gain), ("DetectorMaxThresh", p.max_thresh),
                     ("DetectorMinThresh", p.min_thresh), ("DetectorPlaneFitSize", p.plane_fit_size),
                     ("DetectorPosNegThresh", p.pos_neg_thresh), ("DetectorDoGThresh", p.dog_thresh)],
        "TrackMaper": [("SearchRange", p.search_range), ("QCutOffNumBins", f"{p.qcut_nbins};"),
                       ("QCutOffQuantile", p.qcut_quantile), ("TrackerIterNum", p.tracker_iter_num),
                       ("TrackerInitType", p.tracker_init_type), ("TrackerInitIterNum", p.tracker_init_iter_num),
                       ("TrackerMatchThresh", p.tracker_match_thresh), ("MatchThreshModule", p.match_thresh_module),
                       ("MatchThreshAngle", p.match_thresh_angle), ("MatchNumThresh", p.match_num_thresh),
                       ("ReweigthDistance", p.reweight_distance), ("RegularizeThresh", p.regularize_thresh),
                       ("LocationUncertaintyMatch", p.loc_unc_match), ("ReshapeQAbsolute", p.reshape_q_abs),
                       ("ReshapeQRelative", p.reshape_q_rel), ("LocationUncertainty", p.loc_unc),
                       ("DoReScaling", p.do_rescaling), ("GlobalMatchThreshold", p.global_match_threshold)],
        "Camera": [("CameraDevice", "/dev/video0"), ("ZfX", p.zfx), ("ZfY", p.zfy), ("PPx", p.ppx), ("PPy", p.ppy),
                   ("ImageWidth", p.w), ("ImageHeight", p.h), ("FPS", p.config_fps), ("KcR2", p.kc[0]), ("KcR4", p.kc[1]),
                   ("KcR6", p.kc[2]), ("KcP1", p.kc[3]), ("KcP2", p.kc[4]), ("UseUndistort", getattr(p, "use_undistort", 0)),
                   ("Rotate180", 0)],
        "REBVO": [("CameraType", camera_type), ("VideoNetEnabled", 0), ("SaveLog", save_log), ("LogFile", log_file),
                  ("TrayFile", tray_file), ("TrackKeyFrames", 0), ("StereoAvaiable", 0)],
        "IMU": [("ImuMode", 0)],
    }
    if imu is not None:       # the &IMU section of app/rebvorun/GlobalConfig_EuRoC
        keys = dict(TimeDesinc=0, InitBias=1, InitBiasFrameNum=10, BiasHintX=0.0188, BiasHintY=0.0037, BiasHintZ=0.0776,
                    GiroMeasStdDev=1.6968e-04, GiroBiasStdDev=1.9393e-05, AcelMeasStdDev=2.0000e-3, g_module=9.8,
                    g_module_uncer=0.2e3, g_uncert=2e-3, VBiasStdDev=1e-7, ScaleStdDevMult=1e-2, ScaleStdDevMax=1e-4,
                    ScaleStdDevInit=1.2e-3, CircBufferSize=1000, SampleTime=0.00125)
        keys.update({k: v for k, v in imu.items() if k not in ("mode", "file", "se3", "time_scale")})
        sec["IMU"] = [("ImuMode", imu["mode"])]
        if "file" in imu:
            sec["IMU"] += [("ImuFile", imu["file"]), ("TimeScale", imu.get("time_scale", 1))]
        if "se3" in imu:
            sec["IMU"].append(("CamImuSE3File", imu["se3"]))
        sec["IMU"] += list(keys.items())
    if stereo is not None:
        sec["REBVO"] = [(k, (1 if k == "StereoAvaiable" else v)) for k, v in sec["REBVO"]]
        sec["Stereo"] = [("ZfX", stereo["zfx"]), ("ZfY", stereo["zfy"]), ("PPx", stereo["ppx"]), ("PPy", stereo["ppy"]),
                         ("KcR2", 0), ("KcR4", 0), ("KcR6", 0), ("KcP1", 0), ("KcP2", 0)]
    if dataset is not None:   # (DataSetDir, DataSetFile, TimeScale)
        sec["DataSetCamera"] = [("DataSetDir", dataset[0]), ("DataSetFile", dataset[1]), ("TimeScale", dataset[2])]
        if stereo is not None:
            sec["DataSetCamera"] += [("DataSetDirStereo", stereo["dir"]), ("DataSetFileStereo", stereo["file"])]
    with open(path, "w") as f:
        f.write("// generated by tests/helpers.py\n")
        for name, items in sec.items():
            f.write(f"&{name}   // section\n")
            for k, v in items:
                if f"{name}/{k}" in drop:
                    continue
                f.write(f"    {k}={v!r}" .replace("'", "") + "    //comment\n")


def require_ref():
    """GPU parity tests need oracle/_ref (the reference compiled in place; it travels to the GPU box prebuilt).  Its absence
    is a broken snapshot, not a reason to go green by skipping."""
    import pytest
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref/libreforacle.so is missing: run `make -C oracle` where the reference tree is present (the GPU box "
                    "receives the prebuilt library with the snapshot)")
    return oracle


def tri(k, n):
    p = 2 * (n - 1)
    k = k % p
    return k if k < n else p - k


def hetero_sequence(s, w=752, h=480, scenes=6, pool=12):
    """Sequence `s` of bench.py's heterogeneous batch: scene s % 6 (its own texture and trajectory), started at phase
    (s // 6) of the scene's 12-frame pool.  Returns frame_of(k)."""
    from rebvo_amd import edgehip, synth
    p = edgehip.euroc_params(w, h)
    intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, pool, seed=101 + 7 * (s % scenes), traj_seed=29 + (s % scenes), **intr)]
    ph = (s // scenes) % (2 * (pool - 1))
    return lambda k: frames[tri(k + ph, pool)]


def depths_agree(kg, kr, same):
    """Depth maps of the device (kg) and the reference (kr) on the KeyLines with identical matches: |d rho| <= 1e-5 |rho| + 1e-7 +
    1e-5 s_rho.  The last term: the EKF's gain for a KeyLine that knows nothing about its depth (s_rho of the order of rho or above) is
    close to 1 and its measurement divides by u . (V_xy zf - V_z q0) (edge_tracker.cpp:978-1003), small when the edge runs along the
    epipolar line — a velocity that differs by 1e-9 of the step moves such a depth by 1e-5 of its value and 1e-5 of its own sigma
    (one KeyLine of 16 000 at 1280 x 720, tools/experiments/exp_pipeline_closeness.py)."""
    d = np.abs(kg["rho"][same] - kr["rho"][same])
    return bool(np.all(d <= 1e-5 * np.abs(kr["rho"][same]) + 1e-7 + 1e-5 * kr["s_rho"][same]))
