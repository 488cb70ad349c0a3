"""IMU branch, host side, CPU only: the restated ImuGrabber / BiasCorrect / ScaleEstimator
of rebvo_amd/host (through its flat C view, rebvo/imu_c.h) against the reference's own code in oracle/_ref.

Tolerances: everything that does not go through an SVD is the same sequence of fp64 operations -> 1e-12 relative or
bit-exact where stated; estKaGMEKBias runs 20 Gauss-Newton steps whose linear solve is LAPACK dgesvd_ in the reference
and a Jacobi eigen-solve here -> 1e-7 relative on the state, 1e-4 (in correlation units) on its ill-conditioned covariance.

The reference keeps the EstAcelLsq4 / MeanAcel4 histories in function-local statics, so that comparison runs in a
fresh process (tests/imu_ref_runner.py)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "rebvo_amd", "lib", "librebvohost.so")
REF = os.path.join(ROOT, "oracle", "_ref", "libreforacle.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(HOST) and os.path.exists(REF)),
                                reason="librebvohost.so / oracle/_ref not built")


class Integrated(C.Structure):
    _fields_ = [("n", C.c_int), ("pad", C.c_int), ("dt", C.c_double), ("Rot", C.c_double * 9), ("giro", C.c_double * 3),
                ("acel", C.c_double * 3), ("comp", C.c_double * 3), ("dgiro", C.c_double * 3), ("cacel", C.c_double * 3)]

    def vec(self):
        return np.concatenate([[self.n, self.dt], self.Rot, self.giro, self.acel, self.comp, self.dgiro, self.cacel])


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def so3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


@pytest.fixture(scope="module")
def libs():
    host, ref = C.CDLL(HOST), C.CDLL(REF, mode=C.RTLD_GLOBAL)   # MKL resolves its kernels by dlopen
    for lib, pre in ((host, "rebvo_"), (ref, "ref_")):
        for n in ("imu_grabber_new", "imu_grabber_load"):
            getattr(lib, pre + n).restype = C.c_void_p
        getattr(lib, pre + "imu_grabber_tsample").restype = C.c_double
        getattr(lib, pre + "est_ka_gmek_bias").restype = C.c_double
    host.rebvo_scale_estimator_new.restype = C.c_void_p
    return host, ref


def spd(rng, n, scale=1.0):
    a = rng.normal(size=(n, n))
    return (a @ a.T + n * np.eye(n)) * scale


def test_bias_correct(libs):
    host, ref = libs
    rng = np.random.default_rng(5)
    for it in range(20):
        X = rng.normal(size=6) * 1e-2
        Wx = spd(rng, 6, 1e4)
        Gb = rng.normal(size=3) * 1e-3
        Wb = spd(rng, 3, 1e6)
        Rg = np.eye(3) * (1.7e-4 * 0.05) ** 2
        Rb = np.eye(3) * (1.9e-5 * 0.05) ** 2
        outs = []
        for fn in (host.rebvo_imu_bias_correct, ref.ref_imu_bias_correct):
            x, wx, gb, wb = X.copy(), Wx.copy(), Gb.copy(), Wb.copy()
            fn(dp(x), dp(wx), dp(gb), dp(wb), dp(Rg), dp(Rb))
            outs.append(np.concatenate([x, wx.ravel(), gb, wb.ravel()]))
        assert np.array_equal(outs[0], outs[1]), np.abs(outs[0] - outs[1]).max()   # same operations, same bits


def test_scale_filter_matches_reference(libs):
    host, ref = libs
    rng = np.random.default_rng(11)
    for it in range(12):
        a = 0.3 + 0.5 * rng.random()
        g = so3(rng.normal(size=3) * 0.3) @ np.array([0, 9.8, 0.0])
        a_v = rng.normal(size=3) * 0.3
        a_s = np.tan(a) * a_v - g + rng.normal(size=3) * 1e-2   # (a_s + g) cos a = a_v sin a
        Rot = np.ascontiguousarray(so3(rng.normal(size=3) * 0.02))
        X = np.concatenate([[a + 0.05 * rng.normal()], Rot @ g + rng.normal(size=3) * 0.05, rng.normal(size=3) * 1e-3])
        P = np.diag([1.2e-3 ** 2, 1e-2, 1e-2, 1e-2, 1e-13, 1e-13, 1e-13]) * (1 + rng.random())
        Qg, Qrot, Qbias = np.eye(3) * 4e-6, spd(rng, 3, 1e-7), np.eye(3) * 1e-14
        Rs, Rf = np.eye(3) * 4e-6, spd(rng, 3, 1e-3)
        Wvw = spd(rng, 6, 1e5)
        Xvw = rng.normal(size=6) * 1e-2
        outs = []
        for fn in (host.rebvo_est_ka_gmek_bias, ref.ref_est_ka_gmek_bias):
            x, p, ge, be, xv = X.copy(), P.copy(), np.zeros(3), np.zeros(3), Xvw.copy()
            k = fn(dp(a_s), dp(a_v), C.c_double(1.0), dp(Rot), dp(x), dp(p), dp(Qg), dp(Qrot), dp(Qbias), C.c_double(5e-6),
                   C.c_double(0.2e3 ** 2), dp(Rs), dp(Rf), dp(ge), dp(be), dp(Wvw), dp(xv), C.c_double(9.8))
            outs.append((k, x, p, ge, be, xv))
        (k0, x0, p0, g0, b0, v0), (k1, x1, p1, g1, b1, v1) = outs
        assert abs(k0 - k1) <= 1e-7 * abs(k1) + 1e-12
        assert np.allclose(x0, x1, rtol=1e-7, atol=1e-10)
        assert np.allclose(g0, g1, rtol=1e-7, atol=1e-9) and np.allclose(b0, b1, rtol=1e-6, atol=1e-10)
        assert np.allclose(v0, v1, rtol=1e-7, atol=1e-10)
        # P = inverse of normal equations with a condition number of ~1e11 (bias variances 1e-13 next to 1e-2):
        # compare it in correlation units
        sc = np.sqrt(np.outer(np.diag(p1), np.diag(p1)))
        assert (np.abs(p0 - p1) / sc).max() < 1e-4
        assert np.isfinite(k0) and k0 > 0


def test_problem_without_structural_zeros_equals_the_dense_one(libs):
    """problem_KaGMEKBias (the filters' version: block structure of P, W, dW/da and dF/dx used) against problem_KaGMEKBias_dense
    (the reference's function as written): the same sums over the same non-zero terms in the same order, hence the same bits."""
    host, _ = libs
    rng = np.random.default_rng(3)
    for it in range(40):
        a = rng.uniform(-3, 3)
        x = np.concatenate([[a], rng.normal(size=3) * 9.8, rng.normal(size=3) * 2e-2])
        x_p = x + np.concatenate([[rng.normal() * (4.0 if it % 5 == 0 else 0.05)], rng.normal(size=3) * 0.1, rng.normal(size=3) * 1e-3])
        a_v, a_s = rng.normal(size=3), rng.normal(size=3) + np.array([0, -9.8, 0])
        Rv, Rs, Pp = spd(rng, 3, 1e-3), spd(rng, 3, 4e-6), spd(rng, 7, 1e-2)
        Jd, Fd, Js, Fs = np.zeros(49), np.zeros(7), np.zeros(49), np.zeros(7)
        host.rebvo_problem_ka_gmek_bias(dp(x), dp(a_v), dp(a_s), C.c_double(9.8), dp(x_p), dp(Rv), dp(Rs), C.c_double(0.2e3 ** 2),
                                        dp(Pp), dp(Jd), dp(Fd), dp(Js), dp(Fs))
        assert np.all(np.isfinite(Jd)) and np.abs(Jd).max() > 1
        assert np.array_equal(Jd, Js), (it, np.abs(Jd - Js).max())
        assert np.array_equal(Fd, Fs), (it, np.abs(Fd - Fs).max())


def _imu_rows(rng, n, t0=1.0, dt=0.005):
    t = t0 + dt * np.arange(n)
    g = rng.normal(size=(n, 3)) * 0.2
    a = rng.normal(size=(n, 3)) + np.array([0, 9.8, 0])
    return np.column_stack([t, g, a])


def test_grabber_dataset(libs, tmp_path):
    """LoadDataSet + GrabAndIntegrate over a csv with comments / blank lines, with and without a Cam-IMU transform."""
    host, ref = libs
    rng = np.random.default_rng(2)
    rows = _imu_rows(rng, 400)
    csv = tmp_path / "imu.csv"
    with open(csv, "w") as f:
        f.write("#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z\n\n")
        for r in rows:
            f.write("  %d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n" % (round(r[0] * 1e9), *r[1:]))
    se3 = tmp_path / "se3.csv"
    Rci, Tci = so3(np.array([0.3, -0.2, 0.1])), np.array([0.05, -0.02, 0.01])
    with open(se3, "w") as f:
        for i in range(3):
            f.write(",".join("%.17g" % v for v in Rci[i]) + ",%.17g,\n" % Tci[i])
    for use_se3 in (False, True):
        gh = C.c_void_p(host.rebvo_imu_grabber_load(str(csv).encode(), C.c_double(1e-9)))
        gr = C.c_void_p(ref.ref_imu_grabber_load(str(csv).encode(), C.c_double(1e-9)))
        assert gh.value and gr.value
        assert host.rebvo_imu_grabber_tsample(gh) == ref.ref_imu_grabber_tsample(gr)
        if use_se3:
            assert host.rebvo_imu_grabber_load_se3(gh, str(se3).encode()) == 1
            assert ref.ref_imu_grabber_load_se3(gr, str(se3).encode()) == 1
        t_prev = 0.0
        for k in range(30):   # frame times, the last ones run past the end of the data (empty grabs)
            t = 1.0 + 0.0731 * (k + 1)
            a, b = Integrated(), Integrated()
            host.rebvo_imu_grabber_grab(gh, C.c_double(t_prev), C.c_double(t), C.byref(a))
            ref.ref_imu_grabber_grab(gr, C.c_double(t_prev), C.c_double(t), C.byref(b))
            assert a.n == b.n
            assert np.array_equal(a.vec(), b.vec()), (k, np.abs(a.vec() - b.vec()).max())
            t_prev = t
        assert a.n == 0   # ran off the end
        host.rebvo_imu_grabber_free(gh)
        ref.ref_imu_grabber_free(gr)
    assert not host.rebvo_imu_grabber_load(b"/nonexistent/imu.csv", C.c_double(1.0))
    g = C.c_void_p(host.rebvo_imu_grabber_new(4, C.c_double(0.01)))
    assert host.rebvo_imu_grabber_load_se3(g, b"/nonexistent/se3.csv") == 0
    host.rebvo_imu_grabber_free(g)


def test_grabber_push_ring_and_overflow(libs):
    """ImuMode 1: samples pushed into the ring while frames consume them; a full ring raises (here: -1)."""
    host, ref = libs
    rng = np.random.default_rng(9)
    rows = _imu_rows(rng, 200, t0=0.0, dt=0.01)
    gh = C.c_void_p(host.rebvo_imu_grabber_new(16, C.c_double(0.01)))
    gr = C.c_void_p(ref.ref_imu_grabber_new(16, C.c_double(0.01)))
    pushed, t_prev = 0, 0.0
    for k in range(18):
        for _ in range(8):   # 8 samples per frame interval: the ring of 16 wraps many times
            r = rows[pushed]
            rh = host.rebvo_imu_grabber_push(gh, C.c_double(r[0]), dp(np.ascontiguousarray(r[1:4])), dp(np.ascontiguousarray(r[4:7])))
            rr = ref.ref_imu_grabber_push(gr, C.c_double(r[0]), dp(np.ascontiguousarray(r[1:4])), dp(np.ascontiguousarray(r[4:7])))
            assert rh == rr == 1
            pushed += 1
        t = rows[pushed - 2][0] + 0.001
        a, b = Integrated(), Integrated()
        host.rebvo_imu_grabber_grab(gh, C.c_double(t_prev), C.c_double(t), C.byref(a))
        ref.ref_imu_grabber_grab(gr, C.c_double(t_prev), C.c_double(t), C.byref(b))
        assert a.n == b.n and a.n > 0
        assert np.array_equal(a.vec(), b.vec())
        t_prev = t
    res = []
    for lib, g, pre in ((host, gh, "rebvo_"), (ref, gr, "ref_")):
        out = []
        for i in range(40):   # no consumer any more: the ring fills up and then refuses
            r = rows[(pushed + i) % len(rows)]
            out.append(getattr(lib, pre + "imu_grabber_push")(g, C.c_double(r[0]), dp(np.ascontiguousarray(r[1:4])),
                                                              dp(np.ascontiguousarray(r[4:7]))))
        res.append(out)
    assert res[0] == res[1] and res[0][-1] == -1 and res[0][0] == 1


def test_acel_histories_fresh_process():
    """EstAcelLsq4 / MeanAcel4 against the reference's process-wide statics, in a process of their own (bit-exact)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "imu_ref_runner.py"), "acel"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["calls"] == 40 and out["max_abs_diff_lsq"] == 0.0 and out["max_abs_diff_mean"] == 0.0
    assert out["nonzero"]
