"""Deterministic synthetic image sequences (no dataset is available offline).

Two scenes, both seeded and integer-valued RGB24 (r=g=b), as SURVEY.md §8(d) asks:

* ``rects_sequence``   S1: axis-aligned grey rectangles on a flat background, whole scene translated by a
  sub-pixel-free integer shift per frame.  Cheap; exercises stage A and the matcher.
* ``billboard_sequence`` S2-like: front-parallel grey quads at different depths seen through a pinhole
  camera that follows a smooth SE(3) trajectory -> real parallax, known ground-truth motion.

numpy only; used by tests, bench.py and the replay tool.
"""
import numpy as np


def rects_scene(w, h, n_rects=220, seed=7, pad=64):
    """Big canvas (h+2*pad, w+2*pad) uint8 from which translated crops are taken."""
    rs = np.random.RandomState(seed)
    H, W = h + 2 * pad, w + 2 * pad
    img = np.full((H, W), 40, np.uint8)
    for _ in range(n_rects):
        x0 = int(rs.uniform(0, W))
        y0 = int(rs.uniform(0, H))
        rw = int(10 + rs.uniform(0, 120 * w / 752.0))
        rh = int(10 + rs.uniform(0, 90 * h / 480.0))
        g = int(rs.uniform(0, 255))
        img[y0:y0 + rh, x0:x0 + rw] = g
    return img


def rects_sequence(w, h, n_frames, seed=7, shift=(1.5, 0.7), n_rects=220):
    """Yield RGB24 (h,w,3) uint8 frames: frame k = scene translated by round(k*shift)."""
    pad = int(max(abs(shift[0]), abs(shift[1])) * n_frames) + 8
    canvas = rects_scene(w, h, n_rects, seed, pad)
    for k in range(n_frames):
        dx, dy = int(round(k * shift[0])), int(round(k * shift[1]))
        g = canvas[pad + dy:pad + dy + h, pad + dx:pad + dx + w]
        yield np.repeat(g[:, :, None], 3, axis=2).copy()


def _so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-9:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


class BillboardScene:
    """Front-parallel quads in 3-D + a smooth camera trajectory; renders anti-aliased grey frames."""

    def __init__(self, w, h, fx, fy, cx, cy, n_quads=220, seed=11, ss=2):
        self.w, self.h, self.fx, self.fy, self.cx, self.cy, self.ss = w, h, fx, fy, cx, cy, ss
        rs = np.random.RandomState(seed)
        Z = rs.uniform(1.5, 6.0, n_quads)
        # centre uniformly over a slightly enlarged field of view at depth Z
        u = rs.uniform(-0.15 * w, 1.15 * w, n_quads)
        v = rs.uniform(-0.15 * h, 1.15 * h, n_quads)
        X = (u - cx) / fx * Z
        Y = (v - cy) / fy * Z
        sw = (10 + rs.uniform(0, 120, n_quads)) * (w / 752.0) / fx * Z
        sh = (10 + rs.uniform(0, 90, n_quads)) * (h / 480.0) / fy * Z
        g = rs.randint(0, 256, n_quads)
        order = np.argsort(-Z)  # far first
        self.quads = []
        for i in order:
            c = np.array([[X[i] - sw[i] / 2, Y[i] - sh[i] / 2, Z[i]],
                          [X[i] + sw[i] / 2, Y[i] - sh[i] / 2, Z[i]],
                          [X[i] + sw[i] / 2, Y[i] + sh[i] / 2, Z[i]],
                          [X[i] - sw[i] / 2, Y[i] + sh[i] / 2, Z[i]]])
            self.quads.append((c, int(g[i])))
        self.bg = 40

    def render(self, R, t):
        """Camera pose: X_cam = R @ X_world + t.  Returns (h,w) uint8."""
        ss, w, h = self.ss, self.w, self.h
        W, H = w * ss, h * ss
        img = np.full((H, W), float(self.bg), np.float32)
        for c, g in self.quads:
            pc = c @ R.T + t
            if np.any(pc[:, 2] < 0.2):
                continue
            px = (self.fx * pc[:, 0] / pc[:, 2] + self.cx) * ss + (ss - 1) / 2.0
            py = (self.fy * pc[:, 1] / pc[:, 2] + self.cy) * ss + (ss - 1) / 2.0
            x0, x1 = int(np.floor(px.min())), int(np.ceil(px.max())) + 1
            y0, y1 = int(np.floor(py.min())), int(np.ceil(py.max())) + 1
            x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, W), min(y1, H)
            if x0 >= x1 or y0 >= y1:
                continue
            yy, xx = np.mgrid[y0:y1, x0:x1]
            inside = np.ones(yy.shape, bool)
            for k in range(4):
                ax, ay = px[k], py[k]
                bx, by = px[(k + 1) % 4], py[(k + 1) % 4]
                inside &= ((bx - ax) * (yy - ay) - (by - ay) * (xx - ax)) >= 0
            img[y0:y1, x0:x1][inside] = g
        img = img.reshape(h, ss, w, ss).mean(axis=(1, 3))
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def smooth_trajectory(n_frames, seed=13, v_max=0.02, w_max=np.deg2rad(0.5)):
    """Per-frame twists (v,w) of bounded magnitude, low-pass filtered random walk."""
    rs = np.random.RandomState(seed)
    tw = rs.normal(size=(n_frames + 32, 6))
    k = np.hanning(33)
    k /= k.sum()
    tw = np.stack([np.convolve(tw[:, i], k, mode="valid") for i in range(6)], axis=1)[:n_frames]
    tw /= np.abs(tw).max(axis=0, keepdims=True) + 1e-12
    tw[:, :3] *= v_max
    tw[:, 3:] *= w_max
    tw[:, 0] += 0.6 * v_max  # steady sideways drift so that parallax never vanishes
    return tw


def billboard_sequence(w, h, n_frames, fx=None, fy=None, cx=None, cy=None, seed=11, traj_seed=13, ss=2):
    """Yield (rgb24 frame, R, t) with X_cam_k = R_k X_world + t_k; frame 0 is the identity pose."""
    fx = 458.654 * w / 752.0 if fx is None else fx
    fy = 457.296 * h / 480.0 if fy is None else fy
    cx = 367.215 * w / 752.0 if cx is None else cx
    cy = 248.375 * h / 480.0 if cy is None else cy
    scene = BillboardScene(w, h, fx, fy, cx, cy, seed=seed, ss=ss)
    tw = smooth_trajectory(n_frames, traj_seed)
    R, t = np.eye(3), np.zeros(3)
    for k in range(n_frames):
        g = scene.render(R, t)
        yield np.repeat(g[:, :, None], 3, axis=2).copy(), R.copy(), t.copy()
        dR = _so3_exp(tw[k, 3:])
        R = dR @ R
        t = dR @ t + tw[k, :3]
