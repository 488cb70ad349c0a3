"""A conservative float32 pre-test of build_mask's plane fit (tried in round 3, not kept): how many of the candidates that pass the
two dense gates it rejects for certain, and whether it ever rejects one the exact fp64 fit accepts.  CPU only (reference planes)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle
from rebvo_amd import synth
w, h = 752, 480
op = oracle.euroc_params(w, h)
frames = [f for f, _, _ in synth.billboard_sequence(w, h, 3, seed=11)]
o = oracle.Oracle("ref", op)
for k in range(3):
    _, nav = o.process_frame(frames[k], 0.05 * k)
s = o.cur_slot()
dog, dx, dy = o.plane(s, "dog"), o.plane(s, "dx"), o.plane(s, "dy")
tresh = np.float32(nav.tresh)
gt1 = np.float32(tresh * np.float32(765)); thr_g = gt1 * gt1
gt2 = np.float32(gt1 * np.float32(op.dog_thresh)); thr_d = gt2 * gt2
n2 = dx * dx + dy * dy
gate = np.zeros((h, w), bool); gate[2:h-2, 2:w-2] = ~(n2[2:h-2, 2:w-2] < thr_g)
pos = (dog > 0).astype(np.int32)
from numpy.lib.stride_tricks import sliding_window_view
win = sliding_window_view(dog, (5, 5))          # [h-4, w-4, 5, 5] centred at (y+2, x+2)
npos = sliding_window_view(pos, (5, 5)).sum((2, 3))
pn = 2 * npos - 25
bal = np.abs(pn).astype(np.float64) <= 25 * np.float32(op.pos_neg_thresh)
cand = gate[2:h-2, 2:w-2] & bal
W = win[cand].astype(np.float64)               # [n, 5, 5]
print("candidates", W.shape[0], "kn", nav.kn)
# pseudo inverse: Phi = [x, y, 1] over the window
ys, xs = np.mgrid[-2:3, -2:3]
Phi = np.stack([xs.ravel(), ys.ravel(), np.ones(25)], 1).astype(np.float64)
pinv = np.linalg.inv(Phi.T @ Phi) @ Phi.T
Y = W.reshape(-1, 25)
t = Y @ pinv.T
t0, t1, t2 = t[:, 0], t[:, 1], t[:, 2]
den = t0 * t0 + t1 * t1
with np.errstate(all="ignore"):
    xsv = (-t0 * t2 / den).astype(np.float32); ysv = (-t1 * t2 / den).astype(np.float32)
mx, my = t0.astype(np.float32), t1.astype(np.float32)
acc = ~((np.abs(xsv) > 0.5) | (np.abs(ysv) > 0.5)) & ~((mx * mx + my * my) < thr_d)
print("accepted by exact fit", acc.sum(), "rejected by xs/ys", ((np.abs(xsv) > 0.5) | (np.abs(ysv) > 0.5)).sum(), "by n2m only", (~((np.abs(xsv) > 0.5) | (np.abs(ysv) > 0.5)) & ((mx * mx + my * my) < thr_d)).sum())
# pre-test (float32 emulation, same formulas)
Yf = Y.astype(np.float32)
S = np.abs(Yf).sum(1)
tf = (Yf @ pinv.T.astype(np.float32))
a, b, c = np.abs(tf[:, 0]), np.abs(tf[:, 1]), np.abs(tf[:, 2])
e = np.float32(1e-6 * 1.0001 * np.abs(pinv).max()) * S
n2hi = ((a + e) ** 2 + (b + e) ** 2) * np.float32(1.000001)
cm = c - e
rejx = (cm > 0) & (a > e) & ((a - e) * cm * np.float32(0.999999) > np.float32(0.5001) * n2hi)
rejy = (cm > 0) & (b > e) & ((b - e) * cm * np.float32(0.999999) > np.float32(0.5001) * n2hi)
rejn = n2hi < thr_d * np.float32(0.9999)
rej = rejx | rejy | rejn
print("pre-test rejects", rej.sum(), "of", len(rej), "; wrongly rejected (must be 0):", (rej & acc).sum())
print("|xs| quantiles of rejected-by-exact:", np.nanquantile(np.maximum(np.abs(xsv), np.abs(ysv))[~acc], [0.05, 0.25, 0.5, 0.75]))
