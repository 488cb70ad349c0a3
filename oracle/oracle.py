"""ctypes front-end for the two CPU oracles (TEST INFRASTRUCTURE ONLY — see oracle/README.md).

    Oracle("ref")   -> oracle/_ref/libreforacle.so : the reference's own mtracklib, compiled in place
    Oracle("port")  -> oracle/libedgeport.so       : our plain C++ restatement (oracle/port/)

Both export the ABI in oracle/oracle_abi.h.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (rebvo_amd/) never does.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

KEYLINE_DTYPE = np.dtype(
    [
        ("p_inx", "<i4"),
        ("m_m", "<f4", (2,)),
        ("u_m", "<f4", (2,)),
        ("n_m", "<f4"),
        ("score", "<f4"),
        ("c_p", "<f4", (2,)),
        ("_pad0", "<i4"),
        ("rho", "<f8"),
        ("s_rho", "<f8"),
        ("rho_nr", "<f8"),
        ("s_rho_nr", "<f8"),
        ("rho0", "<f8"),
        ("s_rho0", "<f8"),
        ("p_m", "<f4", (2,)),
        ("p_m_0", "<f4", (2,)),
        ("m_id", "<i4"),
        ("m_id_f", "<i4"),
        ("m_id_kf", "<i4"),
        ("m_num", "<i4"),
        ("m_m0", "<f4", (2,)),
        ("n_m0", "<f8"),
        ("p_id", "<i4"),
        ("n_id", "<i4"),
        ("net_id", "<i4"),
        ("stereo_m_id", "<i4"),
        ("stereo_rho", "<f8"),
        ("stereo_s_rho", "<f8"),
    ]
)
assert KEYLINE_DTYPE.itemsize == 168


class Params(C.Structure):
    """Mirror of OrcParams (oracle_abi.h).  Defaults = app/rebvorun/GlobalConfig_EuRoC + TrackPoints."""

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
        ("tracker_iter_num", C.c_int32), ("tracker_init_type", C.c_int32),
        ("tracker_init_iter_num", C.c_int32),
        ("tracker_match_thresh", C.c_double), ("match_thresh_module", C.c_double),
        ("match_thresh_angle", C.c_double),
        ("match_num_thresh", C.c_uint32), ("do_rescaling", C.c_int32),
        ("reweight_distance", C.c_double), ("regularize_thresh", C.c_double),
        ("loc_unc_match", C.c_double), ("reshape_q_abs", C.c_double),
        ("reshape_q_rel", C.c_double), ("loc_unc", C.c_double),
        ("global_match_threshold", C.c_int32), ("use_undistort", C.c_int32),
        ("config_fps", C.c_double),
    ]


def euroc_params(w=752, h=480, **over):
    """GlobalConfig_EuRoC values (reference app/rebvorun/GlobalConfig_EuRoC:9-52, 61-72), ImuMode=0."""
    p = Params()
    p.w, p.h = w, h
    # principal point / focal scaled with the image when a reduced test size is used
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

class Nav(C.Structure):
    _fields_ = [
        ("t", C.c_double), ("dt", C.c_double),
        ("V", C.c_double * 3), ("W", C.c_double * 3),
        ("P_V", C.c_double * 9), ("P_W", C.c_double * 9),
        ("Rot", C.c_double * 9), ("RotLie", C.c_double * 3), ("Vel", C.c_double * 3),
        ("Pose", C.c_double * 9), ("PoseLie", C.c_double * 3), ("Pos", C.c_double * 3),
        ("Kp", C.c_double), ("RKp", C.c_double), ("s_rho_q", C.c_double),
        ("tresh", C.c_double),
        ("score", C.c_double), ("rel_error", C.c_double), ("rel_error_score", C.c_double),
        ("dtp0", C.c_double), ("dtp1", C.c_double),
        ("retuned_thresh", C.c_float),
        ("kn", C.c_int32), ("klm_fwd", C.c_int32), ("klm_num", C.c_int32), ("kf_matchs", C.c_int32),
        ("estimation_ok", C.c_int32), ("frame", C.c_int32), ("pad0", C.c_int32),
    ]

    def as_dict(self):
        out = {}
        for name, typ in self._fields_:
            v = getattr(self, name)
            out[name] = np.array(v[:]) if hasattr(v, "__len__") else v
        return out


_LIBS = {"ref": os.path.join(HERE, "_ref", "libreforacle.so"), "port": os.path.join(HERE, "libedgeport.so")}


class SeqState(C.Structure):
    """OrcSeqState (oracle_abi.h): the locals of FirstThr / SecondThread that persist from frame to frame."""
    _fields_ = [("tresh", C.c_double), ("t_prev", C.c_double), ("Kp", C.c_double), ("K", C.c_double), ("P_Kp", C.c_double),
                ("V", C.c_double * 3), ("W", C.c_double * 3), ("Pos", C.c_double * 3), ("Pose", C.c_double * 9),
                ("l_kl_num", C.c_int32), ("frame", C.c_int32)]


def half_pixel_keylines(old_kl, field_ikl, ppx, ppy, s_rho_q, w, h):
    """KeyLines of the OLD edge map whose first evaluation in Minimizer_RV is decided by rounding noise.

    init_type 2 always evaluates TryVelRot at X = 0 first (global_tracker.cpp:650-651).  There the re-projected position is
    p_m * z / zf * zf / z = p_m up to a few ulp (ne10wrapper.h:414-424, 433-447), Hom2Img adds the principal point — a *float*
    (cam_model.h:45) — and round2int_positive picks the pixel (global_tracker.cpp:363-364).  A KeyLine detected exactly on a
    half pixel (c_p = n + 0.5: the plane fit's sub-pixel offset is +-0.5, accepted by edge_finder.cpp:146-150) has
    p_m = c_p - pp exactly, so p_m + pp lands exactly on n + 0.5 again and the ulp of noise — which depends on the last bits of
    rho — picks n or n + 1.  When the field holds different KeyLines at the two pixels, the reference's own result for the
    frame depends on those bits: any two implementations whose poses differ in the 16th digit may disagree by one match
    (measured: |dV| ~ 1e-6 on that frame, after which discrete match decisions differ and the trajectories drift apart).

    old_kl: KeyLine records BEFORE the frame is processed; field_ikl: [h, w] KeyLine index plane of the NEW frame's field
    (-1 = empty); s_rho_q: the frame's EstimateQuantile gate.  Returns [(ikl, sorted field entries at the candidate pixels)].
    """
    if len(old_kl) == 0:
        return []
    fx, fy = float(np.float32(ppx)), float(np.float32(ppy))
    px = old_kl["p_m"][:, 0].astype(np.float64) + fx
    py = old_kl["p_m"][:, 1].astype(np.float64) + fy
    hx = (px + 0.5) == np.floor(px + 0.5)
    hy = (py + 0.5) == np.floor(py + 0.5)
    out = []
    for i in np.where((old_kl["s_rho"] <= s_rho_q) & (hx | hy))[0]:
        xs = [int(px[i] + 0.5)] + ([int(px[i] + 0.5) - 1] if hx[i] else [])
        ys = [int(py[i] + 0.5)] + ([int(py[i] + 0.5) - 1] if hy[i] else [])
        vals = {int(field_ikl[y, x]) if (1 <= x < w - 1 and 1 <= y < h - 1) else -2 for x in xs for y in ys}
        if len(vals) > 1:
            out.append((int(i), sorted(vals)))
    return out


SVD_REC = np.dtype([("rows", np.int32), ("cols", np.int32), ("A", np.float64, (6, 6)), ("s", np.float64, 6)])


def available(kind):
    return os.path.exists(_LIBS[kind])


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """One oracle context = one image sequence with the reference's 8-slot PipeBuffer ring."""

    def __init__(self, kind, params, nslots=8):
        if not available(kind):
            raise FileNotFoundError(f"{_LIBS[kind]} missing: run `make -C oracle`")
        self.kind, self.p = kind, params
        self.lib = C.CDLL(_LIBS[kind], mode=C.RTLD_GLOBAL)
        self.w, self.h = params.w, params.h
        L, P = self.lib, kind
        vp, i, d, u = C.c_void_p, C.c_int, C.c_double, C.c_uint
        pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int)

        def fn(name, res, *args):
            f = getattr(L, f"{P}_{name}")
            f.restype, f.argtypes = res, list(args)
            return f

        self._create = fn("create", vp, C.POINTER(Params), i)
        self._destroy = fn("destroy", None, vp)
        self._stage_a = fn("stage_a", i, vp, i, C.c_void_p, pd, pi)
        self._plane = fn("plane", C.POINTER(C.c_float), vp, i, i)
        self._imgc = fn("imgc", C.POINTER(C.c_uint8), vp, i)
        self._undistort_map = fn("undistort_map", i, vp, C.c_void_p, C.c_void_p)
        self._mask = fn("mask", C.POINTER(C.c_int32), vp, i)
        self._kn = fn("kn", i, vp, i)
        self._keylines = fn("keylines", C.c_void_p, vp, i)
        self._retuned = fn("retuned", C.c_float, vp, i)
        self._set_keylines = fn("set_keylines", None, vp, i, C.c_void_p, i, C.c_void_p, C.c_float)
        self._get_fc = fn("get_framecount", u, vp, i)
        self._set_fc = fn("set_framecount", None, vp, i, u)
        self._quantile = fn("quantile", d, vp, i, d, d, d, i)
        self._build_field = fn("build_field", None, vp, i, i, C.c_float)
        self._field = fn("field", C.POINTER(C.c_int32), vp, i)
        self._try_velrot = fn("try_velrot", d, vp, i, i, pd, i, i, d, d, u, d, pd, pd, pd, pd)
        self._minimizer_rv = fn("minimizer_rv", d, vp, i, i, pd, pd, pd, pd, d, i, i, d, pd, pd, d, u, d, pd)
        self._minimizer_v = fn("minimizer_v", d, vp, i, i, pd, pd, d, i, d, u, d, C.c_float)
        self._ext_rot_vel = fn("ext_rot_vel", i, vp, i, pd, d, d, pd, pd, pd)
        self._forward_match = fn("forward_match", i, vp, i, i)
        self._rotate = fn("rotate_keylines", None, vp, i, pd)
        self._directed = fn("directed_matching", i, vp, i, i, pd, pd, pd, pi, d, d, d, d)
        self._regularize = fn("regularize", i, vp, i, d)
        self._ekf = fn("ekf", None, vp, i, pd, pd, pd, d, d, d)
        self._rescale = fn("rescale", d, vp, i, pd, d, u, i)
        self._process = fn("process_frame", i, vp, C.c_void_p, d, C.POINTER(Nav))
        self._cur_slot = fn("cur_slot", i, vp)
        self._run_sequence = fn("run_sequence", i, vp, C.c_void_p, C.c_ulonglong, C.c_void_p, i, d, d, i, C.c_void_p,
                                C.c_void_p) if kind == "ref" else None
        self._minimizer_rv_kf = fn("minimizer_rv_kf", d, vp, i, i, pd, d, d, d, d, i, d, d, u, pd, pi) if kind == "ref" else None
        self._reset = fn("reset_sequence", None, vp)
        self._depth_reset = fn("depth_reset", None, vp)
        self._svd_trace = fn("svd_trace", None, C.c_void_p, i)
        self._svd_trace_count = fn("svd_trace_count", i)
        self._svd_buf = None
        self.ctx = self._create(C.byref(params), nslots)

    # ---- parity diagnostics: the 6x6 decompositions of Minimizer_RV's init phase ------------------------
    def svd_trace_start(self, cap=64):
        """Record the next `cap` decompositions the minimiser asks for (process-wide per oracle library)."""
        self._svd_buf = np.zeros(cap, dtype=SVD_REC)
        self._svd_trace(self._svd_buf.ctypes.data, cap)

    def svd_trace_stop(self):
        """-> structured array (rows, cols, A[6,6], s[6]) in call order."""
        n = self._svd_trace_count()
        self._svd_trace(None, 0)
        out, self._svd_buf = self._svd_buf[:n].copy(), None
        return out

    def seq_state(self):
        f = getattr(self.lib, f"{self.kind}_get_seq_state")
        f.restype, f.argtypes = None, [C.c_void_p, C.POINTER(SeqState)]
        st = SeqState()
        f(self.ctx, C.byref(st))
        return st

    def svd_backsub(self, A, b, chol=False):
        """(_ref only) TooN::SVD<>(A).backsub(b) / TooN::Cholesky<6>(A).backsub(b) as Minimizer_RV calls them."""
        f = self.lib.ref_chol_backsub if chol else self.lib.ref_svd_backsub
        pd = C.POINTER(C.c_double)
        f.restype, f.argtypes = None, [pd, pd, pd]
        A, b, hh = np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64), np.zeros(6)
        f(_dp(A), _dp(b), _dp(hh))
        return hh

    def svd_backend(self, which):
        """(_ref only) 0 = LAPACK dgesvd_ (MKL), 1 = the harness's one-sided Jacobi; returns the previous one."""
        f = self.lib.ref_svd_backend
        f.restype, f.argtypes = C.c_int, [C.c_int]
        return f(which)

    def close(self):
        if self.ctx:
            self._destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stage A -------------------------------------------------------------------------------
    def stage_a(self, slot, rgb, tresh, l_kl_num):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.shape == (self.h, self.w, 3)
        t, l = C.c_double(tresh), C.c_int(l_kl_num)
        kn = self._stage_a(self.ctx, slot, rgb.ctypes.data, C.byref(t), C.byref(l))
        return kn, t.value, l.value

    def plane(self, slot, which):
        idx = {"img0": 0, "img1": 1, "dog": 2, "dx": 3, "dy": 4, "bw": 5}[which]
        p = self._plane(self.ctx, slot, idx)
        return np.ctypeslib.as_array(p, shape=(self.h, self.w)).copy()

    def imgc(self, slot):
        """RGB24 frame that stage A consumed (undistorted when params.use_undistort)."""
        return np.ctypeslib.as_array(self._imgc(self.ctx, slot), shape=(self.h, self.w, 3)).copy()

    def undistort_map(self):
        """image_undistort's map as (inx[h*w,4] with -1 beyond num, iw[h*w,4])."""
        inx = np.empty((self.h * self.w, 4), np.int32)
        iw = np.empty((self.h * self.w, 4), np.int32)
        if self._undistort_map(self.ctx, inx.ctypes.data, iw.ctypes.data) != 0:
            raise RuntimeError("oracle was created without use_undistort")
        return inx, iw

    def mask(self, slot):
        return np.ctypeslib.as_array(self._mask(self.ctx, slot), shape=(self.h, self.w)).copy()

    def kn(self, slot):
        return self._kn(self.ctx, slot)

    def keylines(self, slot):
        kn = self.kn(slot)
        buf = (C.c_char * (168 * kn)).from_address(self._keylines(self.ctx, slot)) if kn else b""
        return np.frombuffer(buf, dtype=KEYLINE_DTYPE, count=kn).copy()

    def retuned(self, slot):
        return self._retuned(self.ctx, slot)

    def set_keylines(self, slot, kl, mask=None, retuned=0.0):
        kl = np.ascontiguousarray(kl, dtype=KEYLINE_DTYPE)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.int32)
        self._set_keylines(self.ctx, slot, kl.ctypes.data, len(kl), None if m is None else m.ctypes.data, retuned)

    def get_framecount(self, slot):
        return self._get_fc(self.ctx, slot)

    def set_framecount(self, slot, fc):
        self._set_fc(self.ctx, slot, fc)

    # ---- stage B -------------------------------------------------------------------------------
    def quantile(self, slot, smin=1e-3, smax=20.0, pct=0.9, n=100):
        return self._quantile(self.ctx, slot, smin, smax, pct, n)

    def build_field(self, slot, radius, min_mod):
        self._build_field(self.ctx, slot, radius, min_mod)

    def field(self, slot):
        return np.ctypeslib.as_array(self._field(self.ctx, slot), shape=(self.h, self.w, 2)).copy()

    def try_velrot(self, slot_new, slot_old, X, reweight, procjf, match_thresh, s_rho_min, match_num_thresh,
                   k_huber, resid_in=None, resid_out=None):
        kn = self.kn(slot_old)
        X = np.ascontiguousarray(X, dtype=np.float64)
        rin = np.zeros(kn) if resid_in is None else np.ascontiguousarray(resid_in, dtype=np.float64)
        rout = np.zeros(kn) if resid_out is None else np.array(resid_out, dtype=np.float64)
        JtJ, JtF = np.zeros((6, 6)), np.zeros(6)
        F = self._try_velrot(self.ctx, slot_new, slot_old, _dp(X), int(reweight), int(procjf), match_thresh,
                             s_rho_min, match_num_thresh, k_huber, _dp(rin), _dp(rout), _dp(JtJ), _dp(JtF))
        return F, JtJ, JtF, rout

    def minimizer_rv(self, slot_new, slot_old, V, W, match_thresh, iter_max, init_type, reweight_distance,
                     max_s_rho, match_num_thresh, init_iter):
        V, W = np.array(V, dtype=np.float64), np.array(W, dtype=np.float64)
        RV, RW, WX = np.eye(3) * 1e50, np.eye(3) * 1e50, np.zeros((6, 6))
        re, res = C.c_double(0), C.c_double(0)
        F = self._minimizer_rv(self.ctx, slot_new, slot_old, _dp(V), _dp(W), _dp(RV), _dp(RW), match_thresh,
                               iter_max, init_type, reweight_distance, C.byref(re), C.byref(res), max_s_rho,
                               match_num_thresh, float(init_iter), _dp(WX))
        return dict(F=F, V=V, W=W, RVel=RV, RW0=RW, W_X=WX, rel_error=re.value, rel_error_score=res.value)

    def minimizer_v(self, slot_new, slot_old, V, match_thresh, iter_max, s_rho_min, match_num_thresh, reweight_distance, min_mod):
        """global_tracker::Minimizer_V<double> (IMU branch) -> dict(F, V, RVel)."""
        V = np.array(V, dtype=np.float64)
        RV = np.zeros((3, 3))
        F = self._minimizer_v(self.ctx, slot_new, slot_old, _dp(V), _dp(RV), match_thresh, iter_max, s_rho_min, match_num_thresh,
                              reweight_distance, min_mod)
        return dict(F=F, V=V, RVel=RV)

    def minimizer_rv_kf(self, slot_kf, slot_cur, X0, Kr, max_s_rho, match_mod, match_ang, rho_tol, iter_max, reweight_distance,
                        match_num_thresh):
        """kfvo::Minimizer_RV_KF<double,false> as kfvo::OptimizePosGT calls it (reference only) -> dict(X, RRV, score_ratio, mnum)."""
        X = np.array(X0, dtype=np.float64)
        RRV = np.zeros((6, 6))
        mnum = C.c_int(0)
        r = self._minimizer_rv_kf(self.ctx, slot_kf, slot_cur, _dp(X), Kr, match_mod, match_ang, rho_tol, iter_max, reweight_distance,
                                  max_s_rho, match_num_thresh, _dp(RRV), C.byref(mnum))
        return dict(X=X, RRV=RRV, score_ratio=r, mnum=mnum.value)

    def ext_rot_vel(self, slot, vel, loc_unc, hub_reweight):
        """edge_tracker::ExtRotVel -> dict(ok, X, Wx, Rx)."""
        vel = np.array(vel, dtype=np.float64)
        X, Wx, Rx = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6))
        ok = self._ext_rot_vel(self.ctx, slot, _dp(vel), loc_unc, hub_reweight, _dp(X), _dp(Wx), _dp(Rx))
        return dict(ok=bool(ok), X=X, Wx=Wx, Rx=Rx)

    # ---- stage C -------------------------------------------------------------------------------
    def forward_match(self, slot_old, slot_new):
        return self._forward_match(self.ctx, slot_old, slot_new)

    def rotate_keylines(self, slot, R):
        R = np.ascontiguousarray(R, dtype=np.float64)
        self._rotate(self.ctx, slot, _dp(R))

    def directed_matching(self, slot_new, slot_old, V, RVel, BackRot, min_thr_mod, min_thr_ang, max_radius, loc_unc):
        V, RVel, BackRot = (np.ascontiguousarray(a, dtype=np.float64) for a in (V, RVel, BackRot))
        kf = C.c_int(0)
        n = self._directed(self.ctx, slot_new, slot_old, _dp(V), _dp(RVel), _dp(BackRot), C.byref(kf),
                           min_thr_mod, min_thr_ang, max_radius, loc_unc)
        return n, kf.value

    def regularize(self, slot, thresh):
        return self._regularize(self.ctx, slot, thresh)

    def ekf(self, slot, V, RVel, RW0, q_abs, q_rel, loc_unc):
        V, RVel, RW0 = (np.ascontiguousarray(a, dtype=np.float64) for a in (V, RVel, RW0))
        self._ekf(self.ctx, slot, _dp(V), _dp(RVel), _dp(RW0), q_abs, q_rel, loc_unc)

    def rescale(self, slot, s_rho_min=20.0, match_num_min=1, re_escale=False):
        rkp = C.c_double(0)
        kp = self._rescale(self.ctx, slot, C.byref(rkp), s_rho_min, match_num_min, int(re_escale))
        return kp, rkp.value

    # ---- whole frame -----------------------------------------------------------------------------
    def process_frame(self, rgb, t):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.shape == (self.h, self.w, 3)
        nav = Nav()
        ran = self._process(self.ctx, rgb.ctypes.data, float(t), C.byref(nav))
        return ran, nav

    def run_sequence(self, pool, idx, t0=0.0, dt=0.05, threads=1, want_navs=False):
        """(_ref only) Replay frames pool[idx[k]] and time them like the reference runs them: threads=1 stage A + B/C back to
        back, threads=2 the reference's FirstThr/SecondThread overlap.  Returns (seconds at which each frame was done, navs)."""
        pool = np.ascontiguousarray(pool, np.uint8)
        idx = np.ascontiguousarray(idx, np.int32)
        done = np.zeros(len(idx), np.float64)
        navs = (Nav * len(idx))() if want_navs else None
        self._run_sequence(self.ctx, pool.ctypes.data, pool[0].nbytes, idx.ctypes.data, len(idx), t0, dt, threads,
                           done.ctypes.data, C.cast(navs, C.c_void_p) if want_navs else None)
        return done, navs

    def cur_slot(self):
        return self._cur_slot(self.ctx)

    # ---- stereo (reference oracle only) ---------------------------------------------------------------
    def set_slot_cam(self, slot, ppx, ppy, zfx, zfy):
        f = self.lib.ref_set_slot_cam
        f.restype, f.argtypes = None, [C.c_void_p, C.c_int] + [C.c_double] * 4
        f(self.ctx, slot, ppx, ppy, zfx, zfy)

    def set_tracker_f32(self, on):
        """Reference oracle only: the tracker's float instantiation, Minimizer_RV<float> (what USE_NE10 builds run)."""
        f = self.lib.ref_set_tracker_f32
        f.restype, f.argtypes = None, [C.c_void_p, C.c_int]
        f(self.ctx, int(on))

    def set_stereo_mode(self, on):
        f = self.lib.ref_set_stereo_mode
        f.restype, f.argtypes = None, [C.c_void_p, C.c_int]
        f(self.ctx, int(on))

    def directed_matching_stereo(self, slot, slot_pair, t, R, min_thr_mod, min_thr_ang, max_radius, loc_unc, q_abs, q_rel, loc_unc_model):
        f = self.lib.ref_directed_matching_stereo
        pd = C.POINTER(C.c_double)
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_int, pd, pd] + [C.c_double] * 7
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        R = np.ascontiguousarray(R, np.float64).reshape(9)
        return f(self.ctx, slot, slot_pair, _dp(t), _dp(R), min_thr_mod, min_thr_ang, max_radius, loc_unc, q_abs, q_rel, loc_unc_model)

    def fuse_stereo_depth(self, slot):
        f = self.lib.ref_fuse_stereo_depth
        f.restype, f.argtypes = None, [C.c_void_p, C.c_int]
        f(self.ctx, slot)

    def enable_stereo(self, ppx, ppy, zfx, zfy, t, R, max_radius=100.0):
        f = self.lib.ref_enable_stereo
        pd = C.POINTER(C.c_double)
        f.restype, f.argtypes = None, [C.c_void_p] + [C.c_double] * 4 + [pd, pd, C.c_double]
        t = np.ascontiguousarray(t, np.float64).reshape(3)
        R = np.ascontiguousarray(R, np.float64).reshape(9)
        f(self.ctx, ppx, ppy, zfx, zfy, _dp(t), _dp(R), max_radius)

    def process_frame_stereo(self, rgb, rgb_pair, t):
        f = self.lib.ref_process_frame_stereo
        f.restype, f.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.POINTER(Nav)]
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        rgb_pair = np.ascontiguousarray(rgb_pair, dtype=np.uint8)
        nav = Nav()
        ran = f(self.ctx, rgb.ctypes.data, rgb_pair.ctypes.data, float(t), C.byref(nav))
        return ran, nav

    def depth_reset(self):
        """REBVO::Reset() semantics (rebvo_second_t.cpp:609-620) applied after the last processed frame."""
        self._depth_reset(self.ctx)

    def reset_sequence(self):
        self._reset(self.ctx)


# ---- IMU branch (reference only; see oracle_abi.h) ----------------------------------------------------------------
class ImuIntegrated(C.Structure):
    """OrcImuIntegrated = rebvo::IntegratedImuData."""
    _fields_ = [("n", C.c_int32), ("pad", C.c_int32), ("dt", C.c_double), ("Rot", C.c_double * 9), ("giro", C.c_double * 3),
                ("acel", C.c_double * 3), ("comp", C.c_double * 3), ("dgiro", C.c_double * 3), ("cacel", C.c_double * 3)]

    def as_row(self):
        return np.concatenate([[self.n, self.dt], self.Rot, self.giro, self.acel, self.comp, self.dgiro, self.cacel])

    @classmethod
    def from_row(cls, r):
        o = cls()
        o.n, o.dt = int(r[0]), float(r[1])
        o.Rot[:] = r[2:11]; o.giro[:] = r[11:14]; o.acel[:] = r[14:17]; o.comp[:] = r[17:20]; o.dgiro[:] = r[20:23]; o.cacel[:] = r[23:26]
        return o


class ImuParams(C.Structure):
    """OrcImuParams; defaults = the &IMU section of app/rebvorun/GlobalConfig_EuRoC."""
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


class NavImu(C.Structure):
    """OrcNavImu."""
    _fields_ = [("Rot", C.c_double * 9), ("RotLie", C.c_double * 3), ("RotGiro", C.c_double * 3), ("Vel", C.c_double * 3),
                ("Pose", C.c_double * 9), ("PoseLie", C.c_double * 3), ("Pos", C.c_double * 3), ("g", C.c_double * 3),
                ("scale", C.c_double), ("dt", C.c_double), ("K", C.c_double), ("Kp", C.c_double), ("RKp", C.c_double),
                ("s_rho_q", C.c_double), ("Vg", C.c_double * 3), ("Bg", C.c_double * 3), ("dVv", C.c_double * 3),
                ("dWv", C.c_double * 3), ("Vgv", C.c_double * 3), ("Vgva", C.c_double * 3), ("Av", C.c_double * 3),
                ("As", C.c_double * 3), ("X", C.c_double * 7), ("b_est", C.c_double * 3), ("u_est", C.c_double * 3),
                ("kn", C.c_int32), ("klm_num", C.c_int32), ("estimation_ok", C.c_int32), ("init", C.c_int32)]


def run_imu_sequence(frames, t, imu_rows, params, imu_params, nslots=8):
    """ref_process_frame_imu over a sequence (frames[k], t[k], imu_rows[k] = ImuIntegrated.as_row()) -> dict of arrays.
    Call it in a fresh process: the reference's acceleration histories are process-wide statics."""
    orc = Oracle("ref", params, nslots)
    L = orc.lib
    L.ref_imu_setup.argtypes = [C.c_void_p, C.POINTER(ImuParams)]
    L.ref_imu_setup.restype = None
    L.ref_process_frame_imu.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(ImuIntegrated), C.POINTER(NavImu)]
    L.ref_process_frame_imu.restype = C.c_int
    L.ref_imu_setup(orc.ctx, C.byref(imu_params))
    out = {name: [] for name, _ in NavImu._fields_}
    out["kl_rho_sum"], out["kl_srho_sum"] = [], []
    out["klprev_n"], out["klprev_rho_sum"], out["klprev_srho_sum"] = [], [], []
    for k in range(len(frames)):
        nav = NavImu()
        f = np.ascontiguousarray(frames[k], np.uint8)
        L.ref_process_frame_imu(orc.ctx, f.ctypes.data, float(t[k]), C.byref(ImuIntegrated.from_row(imu_rows[k])), C.byref(nav))
        for name, ct in NavImu._fields_:
            v = getattr(nav, name)
            out[name].append(np.array(v[:]) if hasattr(v, "__len__") else v)
        kl = orc.keylines(orc.cur_slot())
        out["kl_rho_sum"].append(float(kl["rho"].sum()))
        out["kl_srho_sum"].append(float(kl["s_rho"].sum()))
        # the previous edge map as this frame's tracking left it (what the output callback sees one frame late)
        klp = orc.keylines((k - 1) % nslots) if k >= 1 else kl[:0]
        out["klprev_n"].append(len(klp))
        out["klprev_rho_sum"].append(float(klp["rho"].sum()))
        out["klprev_srho_sum"].append(float(klp["s_rho"].sum()))
    return {k: np.array(v) for k, v in out.items()}
