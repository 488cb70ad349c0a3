"""ctypes binding of libedgehip.so (include/edgehip.h) — the HIP/gfx950 edge pipeline.

This module is only glue: every call goes straight to the C ABI.  There is no CPU fallback; importing
works anywhere (so the CPU test-suite can check the exported symbols) but creating a context needs an
MI355X.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libedgehip.so")

KEYLINE_DTYPE = np.dtype(
    [
        ("p_inx", "<i4"), ("m_m", "<f4", (2,)), ("u_m", "<f4", (2,)), ("n_m", "<f4"), ("score", "<f4"),
        ("c_p", "<f4", (2,)), ("_pad0", "<i4"),
        ("rho", "<f8"), ("s_rho", "<f8"), ("rho_nr", "<f8"), ("s_rho_nr", "<f8"), ("rho0", "<f8"), ("s_rho0", "<f8"),
        ("p_m", "<f4", (2,)), ("p_m_0", "<f4", (2,)),
        ("m_id", "<i4"), ("m_id_f", "<i4"), ("m_id_kf", "<i4"), ("m_num", "<i4"),
        ("m_m0", "<f4", (2,)), ("n_m0", "<f8"),
        ("p_id", "<i4"), ("n_id", "<i4"), ("net_id", "<i4"), ("stereo_m_id", "<i4"),
        ("stereo_rho", "<f8"), ("stereo_s_rho", "<f8"),
    ]
)
assert KEYLINE_DTYPE.itemsize == 168


class Params(C.Structure):
    _fields_ = [
        ("w", C.c_int32), ("h", C.c_int32),
        ("ppx", C.c_double), ("ppy", C.c_double), ("zfx", C.c_double), ("zfy", C.c_double),
        ("kc", C.c_double * 5),
        ("sigma0", C.c_double), ("ksigma", C.c_double),
        ("plane_fit_size", C.c_int32),
        ("pos_neg_thresh", C.c_double), ("dog_thresh", C.c_double),
        ("max_points", C.c_int32), ("reference_points", C.c_int32), ("track_points", C.c_int32),
        ("detector_thresh", C.c_double), ("auto_gain", C.c_double),
        ("max_thresh", C.c_double), ("min_thresh", C.c_double),
        ("search_range", C.c_int32), ("qcut_nbins", C.c_int32),
        ("qcut_quantile", C.c_double),
        ("tracker_iter_num", C.c_int32), ("tracker_init_type", C.c_int32), ("tracker_init_iter_num", C.c_int32),
        ("tracker_match_thresh", C.c_double), ("match_thresh_module", C.c_double), ("match_thresh_angle", C.c_double),
        ("match_num_thresh", C.c_uint32), ("do_rescaling", C.c_int32),
        ("reweight_distance", C.c_double), ("regularize_thresh", C.c_double),
        ("loc_unc_match", C.c_double), ("reshape_q_abs", C.c_double), ("reshape_q_rel", C.c_double),
        ("loc_unc", C.c_double),
        ("global_match_threshold", C.c_int32), ("debug_planes", C.c_int32),
        ("config_fps", C.c_double),
        ("use_undistort", C.c_int32), ("stereo_available", C.c_int32),
    ]


def euroc_params(w=752, h=480, **over):
    """app/rebvorun/GlobalConfig_EuRoC of the reference (+ TrackPoints=12000, ImuMode=0)."""
    p = Params()
    p.w, p.h = w, h
    sx, sy = w / 752.0, h / 480.0
    p.ppx, p.ppy, p.zfx, p.zfy = 367.215 * sx, 248.375 * sy, 458.654 * sx, 457.296 * sy
    p.kc[:] = [-0.28340811, 0.07395907, 0.0, 0.00019359, 1.76187114e-05]
    p.sigma0, p.ksigma = 1.7818, 1.2599
    p.plane_fit_size = 2
    p.pos_neg_thresh, p.dog_thresh = 0.4, 0.095259868922420
    p.max_points, p.reference_points, p.track_points = 16000, 12000, 12000
    p.detector_thresh, p.auto_gain, p.max_thresh, p.min_thresh = 0.01, 5e-7, 0.5, 0.005
    p.search_range, p.qcut_nbins, p.qcut_quantile = 40, 100, 0.9
    p.tracker_iter_num, p.tracker_init_type, p.tracker_init_iter_num = 5, 2, 2
    p.tracker_match_thresh, p.match_thresh_module, p.match_thresh_angle = 0.5, 1.0, 45.0
    p.match_num_thresh, p.do_rescaling = 0, 0
    p.reweight_distance, p.regularize_thresh = 2.0, 0.5
    p.loc_unc_match, p.reshape_q_abs, p.reshape_q_rel, p.loc_unc = 2.0, 1e-4, 1.6968e-04, 1.0
    p.global_match_threshold = 500
    p.debug_planes = 0
    p.config_fps = 20.0
    for k, v in over.items():
        setattr(p, k, v)
    return p


def tum_params(w=640, h=480, **over):
    """app/rebvorun/GlobalConfig_desk.txt of the reference (TUM fr2/desk, ImuMode=0): BASELINE config 4.
    The shipped file has Kc=0/UseUndistort=0; config 4 exercises the undistorter, so callers pass
    use_undistort=1 and a distortion (the EuRoC coefficients by default, SURVEY.md section 8d scene S3)."""
    p = euroc_params(w, h)
    sx, sy = w / 640.0, h / 480.0
    p.ppx, p.ppy, p.zfx, p.zfy = 320.0 * sx, 240.0 * sy, 525.0 * sx, 525.0 * sy
    p.max_points, p.reference_points, p.track_points = 25000, 15000, 12000
    p.detector_thresh, p.auto_gain, p.max_thresh, p.min_thresh = 0.01, 1e-6, 0.05, 0.03
    p.search_range = 20
    p.tracker_iter_num, p.tracker_init_type, p.tracker_init_iter_num = 10, 2, 2
    p.tracker_match_thresh = 1.0
    p.match_num_thresh = 4
    p.reshape_q_rel = 1e-2
    p.config_fps = 50.0
    for k, v in over.items():
        setattr(p, k, v)
    return p

class SeqState(C.Structure):
    _fields_ = [
        ("tresh", C.c_double),
        ("V", C.c_double * 3), ("W", C.c_double * 3),
        ("P_V", C.c_double * 9), ("P_W", C.c_double * 9),
        ("R", C.c_double * 9),
        ("Pose", C.c_double * 9), ("Pos", C.c_double * 3),
        ("Kp", C.c_double), ("P_Kp", C.c_double), ("K", C.c_double),
        ("s_rho_q", C.c_double),
        ("score", C.c_double), ("rel_error", C.c_double), ("rel_error_score", C.c_double),
        ("t_prev", C.c_double), ("dt", C.c_double),
        ("retuned_thresh", C.c_float),
        ("l_kl_num", C.c_int32), ("frame", C.c_int32),
        ("klm_fwd", C.c_int32), ("klm_num", C.c_int32), ("kf_matchs", C.c_int32),
        ("estimation_ok", C.c_int32), ("minimizer_evals", C.c_int32),
    ]


class Nav(C.Structure):
    _fields_ = [
        ("t", C.c_double), ("dt", C.c_double),
        ("V", C.c_double * 3), ("W", C.c_double * 3), ("P_V", C.c_double * 9), ("P_W", C.c_double * 9),
        ("Rot", C.c_double * 9), ("RotLie", C.c_double * 3), ("Vel", C.c_double * 3),
        ("Pose", C.c_double * 9), ("PoseLie", C.c_double * 3), ("Pos", C.c_double * 3),
        ("Kp", C.c_double), ("RKp", C.c_double), ("s_rho_q", C.c_double), ("tresh", C.c_double),
        ("score", C.c_double), ("rel_error", C.c_double), ("rel_error_score", C.c_double),
        ("retuned_thresh", C.c_float),
        ("kn", C.c_int32), ("klm_fwd", C.c_int32), ("klm_num", C.c_int32), ("kf_matchs", C.c_int32),
        ("estimation_ok", C.c_int32), ("frame", C.c_int32), ("minimizer_evals", C.c_int32),
    ]


class KfRequest(C.Structure):
    """edgehip_kf_request (include/edgehip.h)."""
    _fields_ = [("X0", C.c_double * 6), ("Kr", C.c_double), ("max_s_rho", C.c_double)]


class KfResult(C.Structure):
    """edgehip_kf_result (include/edgehip.h)."""
    _fields_ = [("X", C.c_double * 6), ("RRV", C.c_double * 36), ("score_ratio", C.c_double), ("F", C.c_double), ("F0", C.c_double),
                ("evals", C.c_int32), ("mnum", C.c_int32)]


class ImuParams(C.Structure):
    """edgehip_imu_params; defaults = the &IMU section of app/rebvorun/GlobalConfig_EuRoC."""
    _fields_ = [("giro_meas_std", C.c_double), ("giro_bias_std", C.c_double), ("init_bias", C.c_int32),
                ("init_bias_frame_num", C.c_int32), ("bias_init_guess", C.c_double * 3), ("acel_meas_std", C.c_double),
                ("g_module", C.c_double), ("g_module_uncer", C.c_double), ("g_uncert", C.c_double), ("vbias_std", C.c_double),
                ("scale_std_mult", C.c_double), ("scale_std_max", C.c_double), ("scale_std_init", C.c_double)]


def euroc_imu_params(**over):
    p = ImuParams()
    p.giro_meas_std, p.giro_bias_std = 1.6968e-04, 1.9393e-05
    p.init_bias, p.init_bias_frame_num = 1, 10
    p.bias_init_guess[:] = [0.0188, 0.0037, 0.0776]
    p.acel_meas_std, p.g_module, p.g_module_uncer, p.g_uncert, p.vbias_std = 2.0e-3, 9.8, 0.2e3, 2e-3, 1e-7
    p.scale_std_mult, p.scale_std_max, p.scale_std_init = 1e-2, 1e-4, 1.2e-3
    for k, v in over.items():
        if k == "bias_init_guess":
            p.bias_init_guess[:] = v
        else:
            setattr(p, k, v)
    return p


class ImuIntegrated(C.Structure):
    """edgehip_imu_integrated = rebvo::IntegratedImuData."""
    _fields_ = [("n", C.c_int32), ("pad", C.c_int32), ("dt", C.c_double), ("Rot", C.c_double * 9), ("giro", C.c_double * 3),
                ("acel", C.c_double * 3), ("comp", C.c_double * 3), ("dgiro", C.c_double * 3), ("cacel", C.c_double * 3)]

    @classmethod
    def from_row(cls, r):
        """row = [n, dt, Rot(9), giro(3), acel(3), comp(3), dgiro(3), cacel(3)] (oracle.ImuIntegrated.as_row)"""
        o = cls()
        o.n, o.dt = int(r[0]), float(r[1])
        o.Rot[:] = r[2:11]; o.giro[:] = r[11:14]; o.acel[:] = r[14:17]; o.comp[:] = r[17:20]; o.dgiro[:] = r[20:23]; o.cacel[:] = r[23:26]
        return o


class NavImu(C.Structure):
    """edgehip_nav_imu."""
    _fields_ = [("Rot", C.c_double * 9), ("RotLie", C.c_double * 3), ("RotGiro", C.c_double * 3), ("Vel", C.c_double * 3),
                ("Pose", C.c_double * 9), ("PoseLie", C.c_double * 3), ("Pos", C.c_double * 3), ("g", C.c_double * 3),
                ("scale", C.c_double), ("dt", C.c_double), ("K", C.c_double), ("Kp", C.c_double), ("RKp", C.c_double),
                ("s_rho_q", C.c_double), ("Vg", C.c_double * 3), ("Bg", C.c_double * 3), ("dVv", C.c_double * 3),
                ("dWv", C.c_double * 3), ("Vgv", C.c_double * 3), ("Vgva", C.c_double * 3), ("Av", C.c_double * 3),
                ("As", C.c_double * 3), ("X", C.c_double * 7), ("b_est", C.c_double * 3), ("u_est", C.c_double * 3),
                ("kn", C.c_int32), ("klm_num", C.c_int32), ("estimation_ok", C.c_int32), ("init", C.c_int32)]


NAV_DTYPE = np.dtype(Nav)   # numpy view of edgehip_nav (same offsets as the ctypes struct)
assert NAV_DTYPE.itemsize == C.sizeof(Nav)

# every symbol include/edgehip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTS = [
    "edgehip_create", "edgehip_destroy", "edgehip_last_error", "edgehip_abi_version", "edgehip_sync",
    "edgehip_stream", "edgehip_box_widths", "edgehip_upload_rgb", "edgehip_upload_rgb_device",
    "edgehip_stage_a", "edgehip_get_kn", "edgehip_quantile", "edgehip_build_field", "edgehip_try_velrot",
    "edgehip_download_resid", "edgehip_minimizer_rv", "edgehip_forward_match", "edgehip_rotate_keylines",
    "edgehip_directed_matching", "edgehip_regularize_ekf", "edgehip_rescale", "edgehip_process_frame",
    "edgehip_next_slot", "edgehip_cur_slot", "edgehip_read_nav", "edgehip_reset", "edgehip_get_state",
    "edgehip_set_state", "edgehip_get_framecount", "edgehip_set_framecount", "edgehip_download_keylines", "edgehip_download_keylines_batch",
    "edgehip_upload_keylines", "edgehip_download_plane", "edgehip_download_field", "edgehip_profile_enable",
    "edgehip_profile_count", "edgehip_profile_name", "edgehip_profile_read", "edgehip_profile_select",
    "edgehip_upload_rgb_indexed", "edgehip_bind_rgb_indexed", "edgehip_set_nav_log", "edgehip_read_nav_log", "edgehip_read_nav_log_device", "edgehip_read_nav_imu_log", "edgehip_read_stereo_matches_log", "edgehip_set_tracker_precision", "edgehip_export_keylines", "edgehip_export_fetch", "edgehip_export_wait",
    "edgehip_build_undistort_map", "edgehip_download_undistorted", "edgehip_depth_reset", "edgehip_depth_reset_slot", "edgehip_set_slot_camera", "edgehip_directed_matching_stereo",
    "edgehip_alloc_pinned", "edgehip_free_pinned", "edgehip_upload_rgb_pinned", "edgehip_upload_sync", "edgehip_upload_wait", "edgehip_register_host", "edgehip_unregister_host", "edgehip_experiments", "edgehip_fuse_stereo_depth", "edgehip_set_stereo_rig", "edgehip_get_stereo_matches", "edgehip_minimizer_v", "edgehip_ext_rot_vel",
    "edgehip_imu_enable", "edgehip_set_imu", "edgehip_read_nav_imu", "edgehip_minimizer_rv_kf", "edgehip_lm_solve",
    "edgehip_upload_grey8", "edgehip_upload_grey8_pinned", "edgehip_bind_grey8_indexed",
]

_lib = None


def load_library():
    """dlopen libedgehip.so; raises if the HIP extension was not built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C rebvo_amd/csrc` "
                               "(or __graft_entry__.build()); rebvo_amd has no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.edgehip_last_error.restype = C.c_char_p
        _lib.edgehip_profile_name.restype = C.c_char_p
        _lib.edgehip_stream.restype = C.c_void_p
    return _lib


class EdgeHipError(RuntimeError):
    pass


def build_undistort_map(params):
    """Host-only: the bilinear undistortion map edgehip_create() builds for `params`, in the reference's
    undistMapPoint form -> (inx[h*w,4] with -1 beyond num, iw[h*w,4])."""
    lib = load_library()
    n = params.w * params.h
    inx, iw = np.empty((n, 4), np.int32), np.empty((n, 4), np.int32)
    rc = lib.edgehip_build_undistort_map(C.byref(params), C.c_void_p(inx.ctypes.data), C.c_void_p(iw.ctypes.data))
    if rc != 0:
        raise EdgeHipError(f"edgehip_build_undistort_map: {rc}")
    return inx, iw


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class EdgeHip:
    """nseq image sequences advancing in lock-step on one MI355X."""

    def __init__(self, params, nseq=1, nslots=3, device=0):
        self.lib = load_library()
        self.p, self.nseq, self.nslots = params, nseq, nslots
        self.w, self.h, self.cap = params.w, params.h, min(params.max_points, 50000)
        self.ctx = C.c_void_p()
        self._ck(self.lib.edgehip_create(C.byref(params), nseq, nslots, device, C.byref(self.ctx)))

    def _ck(self, rc):
        if rc != 0:
            raise EdgeHipError(f"edgehip error {rc}: {self.lib.edgehip_last_error().decode()}")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.edgehip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def minimizer_v(self, slot_new, slot_old, V, s_rho_min, min_mod, match_thresh, iter_max, match_num_thresh, reweight_distance):
        """global_tracker::Minimizer_V<double> for every sequence -> (V[nseq,3], RVel[nseq,3,3], F[nseq])."""
        V = np.ascontiguousarray(np.broadcast_to(np.asarray(V, np.float64), (self.nseq, 3))).copy()
        smin = np.ascontiguousarray(np.broadcast_to(np.asarray(s_rho_min, np.float64), (self.nseq,))).copy()
        RV, F = np.zeros((self.nseq, 3, 3)), np.zeros(self.nseq)
        self._ck(self.lib.edgehip_minimizer_v(self.ctx, slot_new, slot_old, _dp(V), _dp(smin), C.c_float(min_mod), C.c_double(match_thresh),
                                              iter_max, C.c_uint32(match_num_thresh), C.c_double(reweight_distance), _dp(RV), _dp(F)))
        return V, RV, F

    def minimizer_rv_kf(self, slot_kf, slot_cur, X0, Kr, max_s_rho, match_mod, match_ang, rho_tol, iter_max, reweight_distance,
                        match_num_thresh):
        """kfvo::Minimizer_RV_KF<double,false> for every sequence: the KeyLines of slot_cur against the field of slot_kf's
        KeyLines -> dict(X[nseq,6], RRV[nseq,6,6], score_ratio, F, F0, evals, mnum)."""
        req = (KfRequest * self.nseq)()
        X0 = np.broadcast_to(np.asarray(X0, np.float64), (self.nseq, 6))
        Kr = np.broadcast_to(np.asarray(Kr, np.float64), (self.nseq,))
        ms = np.broadcast_to(np.asarray(max_s_rho, np.float64), (self.nseq,))
        for s in range(self.nseq):
            req[s].X0[:] = list(X0[s])
            req[s].Kr, req[s].max_s_rho = float(Kr[s]), float(ms[s])
        res = (KfResult * self.nseq)()
        self._ck(self.lib.edgehip_minimizer_rv_kf(self.ctx, slot_kf, slot_cur, req, C.c_double(match_mod), C.c_double(match_ang),
                                                  C.c_double(rho_tol), iter_max, C.c_double(reweight_distance),
                                                  C.c_uint32(match_num_thresh), res))
        return dict(X=np.array([list(r.X) for r in res]), RRV=np.array([list(r.RRV) for r in res]).reshape(self.nseq, 6, 6),
                    score_ratio=np.array([r.score_ratio for r in res]), F=np.array([r.F for r in res]), F0=np.array([r.F0 for r in res]),
                    evals=np.array([r.evals for r in res]), mnum=np.array([r.mnum for r in res]))

    def ext_rot_vel(self, slot, vel, loc_unc, hub_reweight):
        """edge_tracker::ExtRotVel for every sequence -> (X[nseq,6], Wx[nseq,6,6], Rx[nseq,6,6], ok[nseq])."""
        vel = np.ascontiguousarray(np.broadcast_to(np.asarray(vel, np.float64), (self.nseq, 3))).copy()
        X, Wx, Rx = np.zeros((self.nseq, 6)), np.zeros((self.nseq, 6, 6)), np.zeros((self.nseq, 6, 6))
        ok = np.zeros(self.nseq, np.int32)
        self._ck(self.lib.edgehip_ext_rot_vel(self.ctx, slot, _dp(vel), C.c_double(loc_unc), C.c_double(hub_reweight), _dp(X), _dp(Wx),
                                              _dp(Rx), C.c_void_p(ok.ctypes.data)))
        return X, Wx, Rx, ok

    def depth_reset(self, seq=-1):
        """REBVO::Reset() (rebvo_second_t.cpp:609-620) for one sequence or all (-1)."""
        self._ck(self.lib.edgehip_depth_reset(self.ctx, seq))

    def depth_reset_slot(self, slot, seq=-1):
        self._ck(self.lib.edgehip_depth_reset_slot(self.ctx, seq, slot))

    def set_slot_camera(self, slot, ppx, ppy, zfx, zfy):
        self._ck(self.lib.edgehip_set_slot_camera(self.ctx, slot, C.c_double(ppx), C.c_double(ppy), C.c_double(zfx), C.c_double(zfy)))

    def directed_matching_stereo(self, slot, slot_pair, t, R, min_thr_mod, min_thr_ang, max_radius, loc_unc, q_abs, q_rel, loc_unc_model):
        """edge_tracker::directed_matching_stereo for every sequence -> nmatch[nseq]."""
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        R = np.ascontiguousarray(R, np.float64).reshape(9)
        nm = np.zeros(self.nseq, np.int32)
        self._ck(self.lib.edgehip_directed_matching_stereo(self.ctx, slot, slot_pair, _dp(t), _dp(R), C.c_double(min_thr_mod),
                                                           C.c_double(min_thr_ang), C.c_double(max_radius), C.c_double(loc_unc),
                                                           C.c_double(q_abs), C.c_double(q_rel), C.c_double(loc_unc_model),
                                                           C.c_void_p(nm.ctypes.data)))
        return nm

    def fuse_stereo_depth(self, slot):
        self._ck(self.lib.edgehip_fuse_stereo_depth(self.ctx, slot))

    def set_stereo_rig(self, slot_pair, t=None, R=None, max_radius=100.0):
        if slot_pair < 0:
            self._ck(self.lib.edgehip_set_stereo_rig(self.ctx, -1, None, None, C.c_double(0)))
            return
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        R = np.ascontiguousarray(R, np.float64).reshape(9)
        self._ck(self.lib.edgehip_set_stereo_rig(self.ctx, slot_pair, _dp(t), _dp(R), C.c_double(max_radius)))

    def get_stereo_matches(self):
        nm = np.zeros(self.nseq, np.int32)
        self._ck(self.lib.edgehip_get_stereo_matches(self.ctx, C.c_void_p(nm.ctypes.data)))
        return nm

    def download_undistorted(self, seq, slot):
        out = np.empty((self.h, self.w, 3), np.uint8)
        self._ck(self.lib.edgehip_download_undistorted(self.ctx, seq, slot, C.c_void_p(out.ctypes.data)))
        return out

    # ---- input ----
    def upload_rgb(self, slot, rgb, seq_first=0):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        if rgb.ndim == 3:
            rgb = rgb[None]
        assert rgb.shape[1:] == (self.h, self.w, 3)
        self._ck(self.lib.edgehip_upload_rgb(self.ctx, slot, rgb.ctypes.data_as(C.c_void_p), seq_first, rgb.shape[0]))

    def upload_grey8(self, slot, grey, seq_first=0):
        """8-bit mono frames [count][h][w] (or one [h][w]): a third of the bytes of upload_rgb, identical results."""
        grey = np.ascontiguousarray(grey, dtype=np.uint8)
        if grey.ndim == 2:
            grey = grey[None]
        assert grey.shape[1:] == (self.h, self.w)
        self._ck(self.lib.edgehip_upload_grey8(self.ctx, slot, grey.ctypes.data_as(C.c_void_p), seq_first, grey.shape[0]))

    def upload_grey8_pinned(self, slot, ptr, seq_first=0, count=None):
        self._ck(self.lib.edgehip_upload_grey8_pinned(self.ctx, slot, ptr, seq_first, self.nseq if count is None else count))

    def bind_grey8_indexed(self, slot, pool_dev_ptr, pool_frames, idx):
        """Stage A of `slot` reads 8-bit mono frame idx[s] of a device pool in place (no copy); the pool needs 16 B of slack."""
        idx = np.ascontiguousarray(idx, np.int32)
        assert idx.shape == (self.nseq,)
        self._ck(self.lib.edgehip_bind_grey8_indexed(self.ctx, slot, C.c_void_p(pool_dev_ptr), pool_frames,
                                                     idx.ctypes.data_as(C.c_void_p)))

    def alloc_pinned_grey8(self, count=None):
        """Page-locked uint8 array [count][h][w] for upload_grey8_pinned (free with free_pinned)."""
        count = self.nseq if count is None else count
        nbytes = count * self.h * self.w
        ptr = C.c_void_p()
        self._ck(self.lib.edgehip_alloc_pinned(C.c_size_t(nbytes), C.byref(ptr)))
        buf = (C.c_uint8 * nbytes).from_address(ptr.value)
        return np.frombuffer(buf, np.uint8).reshape(count, self.h, self.w), ptr

    def upload_rgb_device(self, slot, dev_ptr):
        self._ck(self.lib.edgehip_upload_rgb_device(self.ctx, slot, C.c_void_p(dev_ptr)))

    def bind_rgb_indexed(self, slot, pool_dev_ptr, pool_frames, idx):
        """Stage A of `slot` reads frame idx[s] of a device pool in place (no copy); the pool needs 16 B of slack."""
        idx = np.ascontiguousarray(idx, np.int32)
        assert idx.shape == (self.nseq,)
        self._ck(self.lib.edgehip_bind_rgb_indexed(self.ctx, slot, C.c_void_p(pool_dev_ptr), pool_frames,
                                                   idx.ctypes.data_as(C.c_void_p)))

    def upload_rgb_indexed(self, slot, pool_dev_ptr, pool_frames, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        assert idx.shape == (self.nseq,)
        self._ck(self.lib.edgehip_upload_rgb_indexed(self.ctx, slot, C.c_void_p(pool_dev_ptr), pool_frames,
                                                     idx.ctypes.data_as(C.c_void_p)))

    def set_tracker_precision(self, bits):
        """32: Minimizer_RV<float> / TryVelRot<float> (the reference's USE_NE10 instantiation); 64: the default."""
        self._ck(self.lib.edgehip_set_tracker_precision(self.ctx, bits))

    def set_nav_log(self, length):
        self._ck(self.lib.edgehip_set_nav_log(self.ctx, length))

    def read_nav_log(self, first, count):
        out = (Nav * (count * self.nseq))()
        self._ck(self.lib.edgehip_read_nav_log(self.ctx, first, count, out))
        return [[out[k * self.nseq + s] for s in range(self.nseq)] for k in range(count)]

    def read_nav_log_array(self, first, count):
        """Same records as read_nav_log, as one numpy structured array [count, nseq] (fields = edgehip_nav's): no
        per-record Python objects, for logs of many sequences."""
        out = np.zeros((count, self.nseq), dtype=NAV_DTYPE)
        self._ck(self.lib.edgehip_read_nav_log(self.ctx, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def read_nav_imu_log(self, first, count):
        """The IMU half of the logged records of frames [first, first+count): [count][nseq] NavImu objects."""
        out = (NavImu * (count * self.nseq))()
        self._ck(self.lib.edgehip_read_nav_imu_log(self.ctx, first, count, out))
        return [[out[k * self.nseq + s] for s in range(self.nseq)] for k in range(count)]

    def read_stereo_matches_log(self, first, count):
        """stereo_match_num of the logged frames [first, first+count) (a context with a stereo rig): int32 array [count, nseq]."""
        out = np.zeros((count, self.nseq), dtype=np.int32)
        self._ck(self.lib.edgehip_read_stereo_matches_log(self.ctx, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def read_nav_log_device(self, first, count, out_dev_ptr):
        """Records of frames [first, first+count) as raw edgehip_nav structs into device memory ([count][nseq][NAV_DTYPE.itemsize]
        bytes at out_dev_ptr): what shard.NavMover hands RCCL without a host bounce."""
        self._ck(self.lib.edgehip_read_nav_log_device(self.ctx, first, count, C.c_void_p(out_dev_ptr)))

    def sync(self):
        self._ck(self.lib.edgehip_sync(self.ctx))

    def stream(self):
        return self.lib.edgehip_stream(self.ctx)

    def box_widths(self):
        out = (C.c_int * 6)()
        self._ck(self.lib.edgehip_box_widths(self.ctx, out))
        return list(out)

    # ---- stages ----
    def stage_a(self, slot):
        self._ck(self.lib.edgehip_stage_a(self.ctx, slot))

    def get_kn(self, slot):
        out = np.zeros(self.nseq, np.int32)
        self._ck(self.lib.edgehip_get_kn(self.ctx, slot, out.ctypes.data_as(C.c_void_p)))
        return out

    def quantile(self, slot, smin=1e-3, smax=20.0, pct=0.9, nbins=100):
        self._ck(self.lib.edgehip_quantile(self.ctx, slot, C.c_double(smin), C.c_double(smax), C.c_double(pct), nbins))

    def build_field(self, slot, radius, min_mod=-1.0):
        self._ck(self.lib.edgehip_build_field(self.ctx, slot, radius, C.c_float(min_mod)))

    def try_velrot(self, slot_new, slot_old, X, reweight, procjf, match_thresh, s_rho_min, match_num_thresh, k_huber,
                   resid_in=-1, resid_out=0):
        X = np.ascontiguousarray(np.broadcast_to(np.asarray(X, np.float64), (self.nseq, 6)))
        smin = np.ascontiguousarray(np.broadcast_to(np.asarray(s_rho_min, np.float64), (self.nseq,)))
        out = np.zeros((self.nseq, 43))
        self._ck(self.lib.edgehip_try_velrot(self.ctx, slot_new, slot_old, _dp(X), int(reweight), int(procjf),
                                             C.c_double(match_thresh), _dp(smin), C.c_uint32(match_num_thresh),
                                             C.c_double(k_huber), resid_in, resid_out, _dp(out)))
        return out[:, 42].copy(), out[:, :36].reshape(self.nseq, 6, 6).copy(), out[:, 36:42].copy()

    def download_resid(self, which):
        out = np.zeros((self.nseq, self.cap))
        self._ck(self.lib.edgehip_download_resid(self.ctx, which, _dp(out)))
        return out

    def lm_solve(self, A, b, svd_rule):
        """h = the 6x6 solve of an LM step for every system of A [n, 6, 6], b [n, 6]; svd_rule: the init phase's TooN::SVD<>::backsub."""
        A = np.ascontiguousarray(A, np.float64).reshape(-1, 36)
        b = np.ascontiguousarray(b, np.float64).reshape(-1, 6)
        h = np.zeros_like(b)
        self._ck(self.lib.edgehip_lm_solve(self.ctx, _dp(A), _dp(b), len(A), int(bool(svd_rule)), _dp(h)))
        return h

    def minimizer_rv(self, slot_new, slot_old):
        self._ck(self.lib.edgehip_minimizer_rv(self.ctx, slot_new, slot_old))

    def forward_match(self, slot_old, slot_new):
        self._ck(self.lib.edgehip_forward_match(self.ctx, slot_old, slot_new))

    def rotate_keylines(self, slot, R=None):
        if R is None:
            self._ck(self.lib.edgehip_rotate_keylines(self.ctx, slot, None))
        else:
            R = np.ascontiguousarray(np.broadcast_to(np.asarray(R, np.float64).reshape(-1, 9), (self.nseq, 9)))
            self._ck(self.lib.edgehip_rotate_keylines(self.ctx, slot, _dp(R)))

    def directed_matching(self, slot_new, slot_old):
        self._ck(self.lib.edgehip_directed_matching(self.ctx, slot_new, slot_old))

    def regularize_ekf(self, slot, do_regularize=True, do_ekf=True):
        self._ck(self.lib.edgehip_regularize_ekf(self.ctx, slot, int(do_regularize), int(do_ekf)))

    def rescale(self, slot):
        self._ck(self.lib.edgehip_rescale(self.ctx, slot))

    # ---- whole frame ----
    def next_slot(self):
        return self.lib.edgehip_next_slot(self.ctx)

    def cur_slot(self):
        return self.lib.edgehip_cur_slot(self.ctx)

    def alloc_pinned_frames(self, count=None):
        """Page-locked uint8 array [count][h][w][3] for upload_rgb_pinned (free with free_pinned)."""
        count = self.nseq if count is None else count
        nbytes = count * self.h * self.w * 3
        ptr = C.c_void_p()
        self._ck(self.lib.edgehip_alloc_pinned(C.c_size_t(nbytes), C.byref(ptr)))
        buf = (C.c_uint8 * nbytes).from_address(ptr.value)
        arr = np.frombuffer(buf, np.uint8).reshape(count, self.h, self.w, 3)
        return arr, ptr

    def free_pinned(self, ptr):
        self._ck(self.lib.edgehip_free_pinned(ptr))

    def upload_rgb_pinned(self, slot, ptr, seq_first=0, count=None):
        self._ck(self.lib.edgehip_upload_rgb_pinned(self.ctx, slot, ptr, seq_first, self.nseq if count is None else count))

    def process_frame(self, t):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, np.float64), (self.nseq,)))
        self._ck(self.lib.edgehip_process_frame(self.ctx, _dp(t)))

    def read_nav(self):
        nav = (Nav * self.nseq)()
        self._ck(self.lib.edgehip_read_nav(self.ctx, nav))
        return list(nav)

    # ---- IMU branch on the device ----
    def imu_enable(self, imu_params):
        self._ck(self.lib.edgehip_imu_enable(self.ctx, C.byref(imu_params)))

    def set_imu(self, records):
        """records: one ImuIntegrated per sequence, for the interval that ends with the next process_frame."""
        arr = (ImuIntegrated * self.nseq)(*records)
        self._ck(self.lib.edgehip_set_imu(self.ctx, arr))

    def read_nav_imu(self):
        out = (NavImu * self.nseq)()
        self._ck(self.lib.edgehip_read_nav_imu(self.ctx, out))
        return list(out)

    def reset(self):
        self._ck(self.lib.edgehip_reset(self.ctx))

    # ---- state / data ----
    def get_state(self, seq=0):
        s = SeqState()
        self._ck(self.lib.edgehip_get_state(self.ctx, seq, C.byref(s)))
        return s

    def set_state(self, seq, s):
        self._ck(self.lib.edgehip_set_state(self.ctx, seq, C.byref(s)))

    def get_framecount(self, seq, slot):
        v = C.c_uint32(0)
        self._ck(self.lib.edgehip_get_framecount(self.ctx, seq, slot, C.byref(v)))
        return v.value

    def set_framecount(self, seq, slot, fc):
        self._ck(self.lib.edgehip_set_framecount(self.ctx, seq, slot, C.c_uint32(fc)))

    def download_keylines(self, seq, slot, want_mask=True):
        kl = np.zeros(self.cap, KEYLINE_DTYPE)
        mask = np.zeros((self.h, self.w), np.int32) if want_mask else None
        kn = C.c_int32(0)
        self._ck(self.lib.edgehip_download_keylines(self.ctx, seq, slot, kl.ctypes.data_as(C.c_void_p),
                                                    None if mask is None else mask.ctypes.data_as(C.c_void_p),
                                                    C.byref(kn)))
        return kl[:kn.value].copy(), mask

    def download_keylines_batch(self, slot, seqs, registered=()):
        """AoS KeyLine lists of several sequences of one slot, one packing kernel (edgehip_download_keylines_batch).  `registered`:
        positions in `seqs` whose destination is page-locked first (edgehip_register_host: the copy lands in it directly)."""
        seqs = np.ascontiguousarray(seqs, dtype=np.int32)
        n = len(seqs)
        bufs = [np.zeros(self.cap, KEYLINE_DTYPE) for _ in range(n)]
        for j in registered:
            self._ck(self.lib.edgehip_register_host(C.c_void_p(bufs[j].ctypes.data), C.c_size_t(bufs[j].nbytes)))
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        kn = np.zeros(n, np.int32)
        try:
            self._ck(self.lib.edgehip_download_keylines_batch(self.ctx, slot, n, seqs.ctypes.data_as(C.c_void_p), ptrs, kn.ctypes.data_as(C.c_void_p)))
        finally:
            for j in registered:
                self._ck(self.lib.edgehip_unregister_host(C.c_void_p(bufs[j].ctypes.data)))
        return [b[:k].copy() for b, k in zip(bufs, kn)]

    def export_keylines(self, seqs):
        """edgehip_export_keylines: the OLD slot of the frame processed last, packed in-stream (no synchronisation) -> ticket."""
        seqs = np.ascontiguousarray(seqs, dtype=np.int32)
        t = C.c_int(0)
        self._ck(self.lib.edgehip_export_keylines(self.ctx, len(seqs), seqs.ctypes.data_as(C.c_void_p), C.byref(t)))
        return (t.value, len(seqs))

    def export_fetch(self, ticket, kns, registered=True):
        """Enqueue the copies of kns[j] records per list (does not block); returns the destination arrays, valid after export_wait."""
        tid, n = ticket
        kns = np.ascontiguousarray(kns, dtype=np.int32)
        assert len(kns) == n
        bufs = [np.zeros(self.cap, KEYLINE_DTYPE) for _ in range(n)]
        if registered:
            for b in bufs:
                self._ck(self.lib.edgehip_register_host(C.c_void_p(b.ctypes.data), C.c_size_t(b.nbytes)))
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        self._ck(self.lib.edgehip_export_fetch(self.ctx, tid, kns.ctypes.data_as(C.c_void_p), ptrs))
        return {"bufs": bufs, "kns": kns, "registered": registered, "ticket": ticket}

    def export_wait(self, fetched):
        """Block until the ticket's copies have landed; returns the lists."""
        self._ck(self.lib.edgehip_export_wait(self.ctx, fetched["ticket"][0]))
        if fetched["registered"]:
            for b in fetched["bufs"]:
                self._ck(self.lib.edgehip_unregister_host(C.c_void_p(b.ctypes.data)))
        return [b[:k].copy() for b, k in zip(fetched["bufs"], fetched["kns"])]

    def export_drop(self, ticket):
        self._ck(self.lib.edgehip_export_wait(self.ctx, ticket[0]))

    def upload_keylines(self, seq, slot, kl, mask=None, retuned=0.0):
        kl = np.ascontiguousarray(kl, dtype=KEYLINE_DTYPE)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.int32)
        self._ck(self.lib.edgehip_upload_keylines(self.ctx, seq, slot, kl.ctypes.data_as(C.c_void_p), len(kl),
                                                  None if m is None else m.ctypes.data_as(C.c_void_p),
                                                  C.c_float(retuned)))

    def download_plane(self, seq, which):
        idx = {"img0": 0, "img1": 1, "dog": 2, "dx": 3, "dy": 4}[which]
        out = np.zeros((self.h, self.w), np.float32)
        self._ck(self.lib.edgehip_download_plane(self.ctx, seq, idx, out.ctypes.data_as(C.c_void_p)))
        return out

    def download_field(self, seq):
        out = np.zeros((self.h, self.w, 2), np.int32)
        self._ck(self.lib.edgehip_download_field(self.ctx, seq, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- measurement ----
    def profile_enable(self, on=True):
        self._ck(self.lib.edgehip_profile_enable(self.ctx, int(on)))

    def profile_select(self, names=None):
        n = self.lib.edgehip_profile_count()
        mask = 0
        for i in range(n):
            if names is None or self.lib.edgehip_profile_name(i).decode() in names:
                mask |= 1 << i
        self._ck(self.lib.edgehip_profile_select(self.ctx, C.c_uint64(mask)))

    def profile_read(self):
        n = self.lib.edgehip_profile_count()
        ms = np.zeros(n)
        calls = np.zeros(n, np.int64)
        self._ck(self.lib.edgehip_profile_read(self.ctx, _dp(ms), calls.ctypes.data_as(C.c_void_p)))
        return {self.lib.edgehip_profile_name(i).decode(): (float(ms[i]), int(calls[i])) for i in range(n)}
