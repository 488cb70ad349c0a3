"""GlobalConfig files in the reference's format, written from an edgehip Params struct (tests, bench.py, examples)."""


def write_global_config(path, p, log_file="", tray_file="", save_log=0, camera_type=3, drop=(), dataset=None, imu=None, stereo=None, gpu=None, affinity=None):
    """A GlobalConfig file in the reference's format (app/rebvorun/GlobalConfig_EuRoC) from a Params struct.
    `drop` lists "Section/Key" entries to leave out (missing-key error tests).  `imu` = dict(mode=1|2, file=..., se3=...,
    time_scale=..., plus any key of the &IMU section to override) switches the IMU branch on.  `stereo` = dict(dir=..., file=...,
    ppx=, ppy=, zfx=, zfy=) sets StereoAvaiable with the pair camera's list and the &Stereo intrinsics.  `gpu` = dict(device=,
    group=, size=, mono=, tracker_precision=) writes the optional &GPU section (Device, BatchGroup, BatchSize, MonoUpload, TrackerPrecision: rebvo_amd/host/include/rebvo/rebvo.h).
    `affinity` = (CamaraT1, CamaraT2, CamaraT3) writes &ProcesorConfig with SetAffinity=1."""
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
    if affinity is not None:
        sec["ProcesorConfig"] = [("SetAffinity", 1), ("CamaraT1", affinity[0]), ("CamaraT2", affinity[1]), ("CamaraT3", affinity[2])]
    if gpu is not None:
        sec["GPU"] = [(k_, gpu[g_]) for k_, g_ in (("Device", "device"), ("BatchGroup", "group"), ("BatchSize", "size"), ("MonoUpload", "mono"),
                                                       ("TrackerPrecision", "tracker_precision")) if g_ in gpu]
    with open(path, "w") as f:
        f.write("// generated by rebvo_amd/config.py\n")
        for name, items in sec.items():
            f.write(f"&{name}   // section\n")
            for k, v in items:
                if f"{name}/{k}" in drop:
                    continue
                f.write(f"    {k}={v!r}" .replace("'", "") + "    //comment\n")
