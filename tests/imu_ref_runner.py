"""Helper run as its own process by the IMU tests: the reference keeps filter histories in function-local statics
(ScaleEstimator::EstAcelLsq4 / MeanAcel4, scaleestimator.cpp:42-44, 97), so anything that depends on them needs a
process in which the oracle has not been called before.

    imu_ref_runner.py acel                      -> JSON {calls, max_abs_diff_lsq, max_abs_diff_mean, nonzero}
    imu_ref_runner.py sequence <in.npz> <out.npz>  -> reference IMU-branch sequence (ref_process_frame_imu) on the frames,
                                                   time stamps and integrated IMU data of in.npz (keys frames, t, imu,
                                                   w, h, over / imu_over = JSON parameter overrides)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def so3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def run_acel():
    host = C.CDLL(os.path.join(ROOT, "rebvo_amd", "lib", "librebvohost.so"))
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libreforacle.so"), mode=C.RTLD_GLOBAL)
    host.rebvo_scale_estimator_new.restype = C.c_void_p
    se = C.c_void_p(host.rebvo_scale_estimator_new())
    rng = np.random.default_rng(3)
    a_r, a_h, m_r, m_h = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)
    d1 = d2 = 0.0
    for k in range(40):
        vel, sa = rng.normal(size=3), rng.normal(size=3) + np.array([0, 9.8, 0])
        R = np.ascontiguousarray(so3(rng.normal(size=3) * 0.05))
        dt = 0.05 + 0.001 * (k % 5)
        ref.ref_est_acel_lsq4(dp(vel), dp(a_r), dp(R), C.c_double(dt))
        host.rebvo_est_acel_lsq4(se, dp(vel), dp(a_h), dp(R), C.c_double(dt))
        ref.ref_mean_acel4(dp(sa), dp(m_r), dp(R))
        host.rebvo_mean_acel4(se, dp(sa), dp(m_h), dp(R))
        d1, d2 = max(d1, np.abs(a_r - a_h).max()), max(d2, np.abs(m_r - m_h).max())
    print(json.dumps({"calls": 40, "max_abs_diff_lsq": d1, "max_abs_diff_mean": d2, "nonzero": bool(np.abs(a_r).max() > 0)}))


def run_sequence(inp, outp):
    from oracle import oracle
    d = np.load(inp)
    w, h = int(d["w"]), int(d["h"])
    params = oracle.euroc_params(w, h, **json.loads(str(d["over"])))
    imu_params = oracle.euroc_imu_params(**json.loads(str(d["imu_over"])))
    res = oracle.run_imu_sequence(d["frames"], d["t"], d["imu"], params, imu_params)
    np.savez(outp, **res)


if __name__ == "__main__":
    if sys.argv[1] == "acel":
        run_acel()
    elif sys.argv[1] == "sequence":
        run_sequence(sys.argv[2], sys.argv[3])
    else:
        sys.exit(2)
