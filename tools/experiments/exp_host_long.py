"""Long replay through the C++ host layer (REBVO class, three threads) vs the reference."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rebvo_amd import edgehip, synth
from oracle import oracle
from helpers import write_global_config
EXE = os.path.join(ROOT, "rebvo_amd", "lib", "custom_cam_replay")
w, h, npool, n, t0, dt = 752, 480, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 40, 1.0, 0.05
pool = [f for f, _, _ in synth.billboard_sequence(w, h, npool, seed=11)]
P2 = 2 * (npool - 1)
tri = lambda k: (k % P2) if (k % P2) < npool else P2 - (k % P2)
frames = [pool[tri(k)] for k in range(n)]
tmp = tempfile.mkdtemp()
np.stack(frames).tofile(os.path.join(tmp, "frames.rgb24"))
cfg, dump, tray = os.path.join(tmp, "cfg"), os.path.join(tmp, "dump.txt"), os.path.join(tmp, "tray.txt")
write_global_config(cfg, edgehip.euroc_params(w, h), log_file=os.path.join(tmp, "log.m"), tray_file=tray, save_log=1)
r = subprocess.run([EXE, cfg, os.path.join(tmp, "frames.rgb24"), str(n), str(t0), str(dt), dump], capture_output=True, text=True, timeout=600)
print("rc", r.returncode, r.stdout[-300:], r.stderr[-300:])
rows = np.loadtxt(dump, ndmin=2)
print("rows", rows.shape)
orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
navs, worst = [], 0.0
for k, f in enumerate(frames):
    _, nav = orc.process_frame(f, t0 + dt * k); navs.append(nav)
    if k == 0: continue
    j = k - 1; row = rows[j]
    kl = orc.keylines(j % 8)
    d = 0.0
    if int(row[2]) != len(kl): d = 1.0
    if j > 0:
        d = max(d, np.abs(row[5:8] - np.array(navs[j].Pos[:])).max(), np.abs(row[8:11] - np.array(navs[j].PoseLie[:])).max(),
                float(int(row[4]) != navs[j].estimation_ok))
    d = max(d, abs(row[14] - kl["rho"].sum()) / abs(kl["rho"].sum()))
    if d > 1e-7: print("frame", j, "d", d, int(row[2]), len(kl))
    worst = max(worst, d)
print("worst", worst)
