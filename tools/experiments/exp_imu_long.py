"""The IMU-branch host replays of tests/test_imu_gpu.py with more frames (N from argv)."""
import os, sys, pathlib, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_imu_gpu as T
T.N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
import numpy as np
def _report(rows, o):
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / (np.max(np.abs(np.asarray(b))) + 1e-300))
    for k in range(1, T.N - 1):
        row = rows[k]
        print(k, "klm", int(row[3]), int(o["klm_num"][k]), "Vel %.1e RotLie %.1e Vg %.1e Bg %.1e scale %.1e X %.1e rho_sum %.1e" % (
            rel(row[11:14], o["Vel"][k]), rel(row[16:19], o["RotLie"][k]), rel(row[29:32], o["Vg"][k]), rel(row[32:35], o["Bg"][k]),
            rel(row[25:26], [o["scale"][k]]), rel(row[35:42], o["X"][k]), rel(row[14:15], [o["klprev_rho_sum"][k + 1]])), flush=True)
if os.environ.get("REPORT"):
    T._compare = _report
for fn in (T.test_imu_mode2_replay_matches_reference,):
    d = pathlib.Path(tempfile.mkdtemp())
    try:
        fn(d)
        print(fn.__name__, "N =", T.N, "ok", flush=True)
    except AssertionError as e:
        import traceback; traceback.print_exc()
        print(fn.__name__, "N =", T.N, "FAILED", str(e)[:300], flush=True)
