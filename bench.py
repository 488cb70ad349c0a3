#!/usr/bin/env python3
"""bench.py — frames/sec of the REBVO edge pipeline (DoG + extract + track + depth EKF) on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched
by torch.distributed.run with one rank per GPU.  ONE JSON line on rank 0's stdout, at most 4096 bytes
(`compact_line`); the full record of the run goes to bench_extras.json next to this file, and to stderr.

Workload of the default line (BASELINE.json configs[2] at the EuRoC size of configs[1]): `--nseq` independent synthetic
752x480 sequences per GPU ("billboards": textured quads at different depths seen by a moving pinhole camera, EuRoC
intrinsics and GlobalConfig_EuRoC parameters, ImuMode=0).  One step = one new frame of EVERY sequence through the full
path: RGB->grey, scale space, DoG, KeyLine extraction, distance field, Minimizer_RV (12 TryVelRot evaluations +
device-side LM), forward match, rotate, directed matching, regularise, EKF, rescale, pose integration.  Frames are
resident in HBM before the timed region — every sequence reads its frames from its own copy (`--input distinct`: 26.6 GB
at the defaults, so stage A's input comes out of HBM, not out of a cache-resident pool) — nothing is skipped inside it and
there is no host synchronisation per step.  Sequences shard across GPUs with no data-path collective ("weak" scaling: the
per-GPU work is fixed); the per-frame nav records go to rank 0 over RCCL on a side stream, double buffered, off the timed
critical path.

On the line (rank 0):
  roofline         the kernel group with the most time (HIP events on the stream it runs on, over the timed region): the bytes
                   one launch cannot avoid moving / mean launch duration against 8 TB/s = `frac`;
                   `frac_survey_formula` where SURVEY 8(d)'s per-unit figure counts more; `traffic` / `frac_on_traffic` from the
                   committed rocprofv3 PMC passes of this command, `issue_frac` (vector-ALU-busy share of all SIMD cycles) from the
                   committed SQ passes — both used only when stamped with the sha of the library sources that are running
  cpu_baseline     the reference's own mtracklib (oracle/_ref, compiled in place from the reference sources) on the host
                   cores of this box: one core (value), the reference's own two-thread overlap, node-saturating (`modes`)
  pose_rmse        BASELINE.json's "pose RMSE vs CPU ref": 33 sequences of the batch (N > 1: five per rank, summed over the
                   ranks) against the CPU reference on the same frames, with the free-running parity counts
  single_sequence_ms_per_frame   one sequence per launch: the literal form of configs[2] / [4]
  host_surface     frames/s through the rebvo::REBVO plugin surface (requestCustomCamBuffer -> getNav) for 1 / 8 / 64 objects,
                   the 8 and the 64 as one batch group each (rebvo_amd/host/src/batch_group.cpp), RGB24 crossing PCIe inside
In bench_extras.json only: roofline_kernels (every group: frac, traffic, issue_frac), kernel_us_per_step, batch_sweep, the
per-sequence parity details, and with `--extras` the heterogeneous batch with its teacher-forced replay, the PCIe-inclusive
legs and the other BASELINE configurations (stage_a, tum_undistort, ImuMode=2), each in a process of its own.

`--config stage_a` (BASELINE configs[1]): DoG + edge_finder KeyLine extraction alone, reported as HBM GB/s.
`--config tum_undistort` (BASELINE configs[3]): TUM 640x480, GlobalConfig_desk.txt parameters, undistortion on.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 752, 480
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FRAME_DT = 0.05        # seconds between frames of a replay (EuRoC: 20 Hz); a mounted data set sets its own mean interval


def find_dataset(path):
    """The image list of a mounted data set in the layouts the reference's DataSetCam configs point at (datasetcam.cpp:51-85):
    EuRoC `<path>/mav0/cam0/data.csv` + `data/` (also `<path>/cam0/...` or the cam0 directory itself; nanosecond stamps) or TUM
    `<path>/rgb.txt` (seconds, names relative to <path>).  -> (kind, DataSetDir, DataSetFile, TimeScale) or None."""
    if not path or not os.path.isdir(path):
        return None
    for sub in ("mav0/cam0", "cam0", ""):
        d = os.path.join(path, sub) if sub else path
        if os.path.isfile(os.path.join(d, "data.csv")) and os.path.isdir(os.path.join(d, "data")):
            return "euroc", os.path.join(d, "data") + "/", os.path.join(d, "data.csv"), 1e-9
    if os.path.isfile(os.path.join(path, "rgb.txt")):
        return "tum", path.rstrip("/") + "/", os.path.join(path, "rgb.txt"), 1.0
    return None


def load_dataset(found, w, h, max_frames):
    """The first `max_frames` images of the list through the host library's own DataSetCam (rebvo/dataset_c.h: list parser, time
    stamps, PNG / PGM / PPM decoder — the code dataset_replay feeds the tracker with).  -> (frames [n][h][w][3] u8, stamps [n])."""
    import ctypes as C
    kind, ddir, dfile, scale = found
    lib = C.CDLL(os.path.join(ROOT, "rebvo_amd", "lib", "librebvohost.so"))
    lib.rebvo_dataset_open.restype = C.c_void_p
    lib.rebvo_dataset_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double]
    lib.rebvo_dataset_frames.argtypes = [C.c_void_p]
    lib.rebvo_dataset_grab.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.rebvo_dataset_close.argtypes = [C.c_void_p]
    ds = lib.rebvo_dataset_open(ddir.encode(), dfile.encode(), w, h, scale)
    if not ds:
        raise SystemExit(f"bench.py: --dataset: cannot read the image list {dfile}")
    n = min(int(lib.rebvo_dataset_frames(ds)), max_frames)
    frames = np.zeros((n, h, w, 3), np.uint8)
    ts = np.zeros(n)
    t, mono = C.c_double(), C.c_int()
    for k in range(n):
        if lib.rebvo_dataset_grab(ds, frames[k].ctypes.data, C.byref(t), C.byref(mono)) != 0:
            lib.rebvo_dataset_close(ds)
            raise SystemExit(f"bench.py: --dataset: frame {k} of {dfile} could not be read as a {w}x{h} image "
                             "(EuRoC is 752x480: the default line; TUM 640x480: --config tum_undistort)")
        ts[k] = t.value
    lib.rebvo_dataset_close(ds)
    if n < 8:
        raise SystemExit(f"bench.py: --dataset: {n} frames in {dfile}; at least 8 are needed")
    return frames, ts


def tri(k, n):
    """Triangle wave over [0, n-1]: forward then backward through the frame pool (continuous motion)."""
    p = 2 * (n - 1)
    k = k % p
    return k if k < n else p - k


def tri_v(k, n):
    """tri() of an integer array."""
    p = 2 * (n - 1)
    k = np.asarray(k, dtype=np.int64) % p
    return np.where(k < n, k, p - k)


# Matching in one pass (the library's default for ImuMode 0 without a stereo pair, EDGEHIP_FUSE_MATCH): FordwardMatch's copy of the
# ten fields happens inside k_directed, so its 100 bytes per KeyLine are that group's, and its arbitration (24 bytes) rides on the
# rotate_keylines pass: no C.forward_match group is launched.
ONE_PASS_MATCHING = os.environ.get("EDGEHIP_FUSE_MATCH", "1") != "0"


def algorithmic_bytes(group, kn, n_px, radius, nseq):
    """Compulsory bytes per launch of a kernel group, for `nseq` batched sequences: what THIS launch cannot
    avoid moving.  A TryVelRot evaluation is 84 B per KeyLine: 40 B of the KeyLine's own streams + 44 B
    gathered per evaluation; the two-chain launch (k_try_velrot2, group B.try_velrot2) carries two evaluations but streams the
    KeyLine's 40 B once: 40 + 2 x 44 = 128 B.  survey_bytes() keeps SURVEY 8(d)'s 84 B x evaluations for `frac_survey_formula`."""
    per_seq = {
        # stage A pieces: inputs/outputs each kernel cannot avoid
        "A.rgb_rowscan": 3 * n_px + 4 * n_px,                 # RGB24 in, row-prefix plane out
        "A.colscan": 2 * 4 * n_px,                           # plane in, plane out (per plane)
        "A.avg_rowscan": 2 * 4 * n_px,                       # integral in, integral out (per plane)
        "A.detect": 2 * 4 * n_px + 4 * n_px + 20 * kn,       # two integrals in, mask + candidates out
        # one-pass level kernel, mean of its three launches: (3N in + 4N out) + (4N + 4N) + 2 * (4N + 4N)
        "A.level": (7 * n_px + 8 * n_px + 16 * n_px) / 3.0,
        "A.compact": 20 * kn + 168 * kn,                     # candidates in, KeyLine SoA out
        # the fused stage-A kernel: SURVEY.md 8(d) stage A = RGB24 in + img_mask_kl out + the KeyLine records, of which the
        # detector writes what the plane fit produces (p_inx, {xs, ys, m_m}, p_id: 24 B) and the join kernel that follows
        # the rest (168 - 24 B), after reading those 24 B back and probing three mask neighbours
        "A.fused": 3 * n_px + 4 * n_px + 24 * kn,
        "A.join_retune": (24 + 3 * 4) * kn + (168 - 24) * kn,
        "B.try_velrot": 84 * kn,
        "B.try_velrot2": (40 + 2 * 44) * kn,
        # What the binned builder (k_field_bin + k_field_raster) has to move: every KeyLine's 32-byte record in and the 12 bytes of
        # FordwardMatch's arbitration reset out (k_field_bin), one 4-byte bin entry written and read back plus the record gathered
        # again per (KeyLine, tile) pair — a +-r segment crosses 1 + 2r (|u_x| + |u_y|) / 64 tiles of 64 px, 1 + 2r (4/pi) / 64 on
        # average over directions — and the 2-byte KeyLine-index plane out.  SURVEY 8(d)'s 8N + 8*2r*kn (a cleared 8-byte field and one
        # 8-byte store per sample) describes the reference's scatter; in the tiled form the samples never leave LDS, and pricing the
        # launch on them flattered it 2x: kept as frac_survey_formula.
        "B.build_field": 2 * n_px + (32 + 12) * kn + (4 + 4 + 32) * kn * (1 + 2 * radius * (4 / np.pi) / 64),
        "B.tvr_prepare": 0,   # per-sequence set-up of the minimisation since P0 is rebuilt in registers by k_try_velrot (latency, no stream)
        "B.lm_step": 0,
        "B.quantile": 8 * kn,
        "C.forward_match": (4 + 8 + 8 + 4 + (0 if ONE_PASS_MATCHING else 100)) * kn,
        # p_m, rho, s_rho, m_m in and out (the gather record's copy of m_m is no longer rewritten); one pass: + m_id_f, key, win
        "C.rotate": (2 * (8 + 16 + 8) + (4 + 8 + 8 + 4 if ONE_PASS_MATCHING else 0)) * kn,
        "C.directed_matching": (4 * 40 + 2 * 168 + (100 if ONE_PASS_MATCHING else 0)) * kn,      # SURVEY.md §8(d) (+ the forward copy)
        "C.regularize_ekf": (3 * 16 + 16 + 100) * kn,
        # SURVEY 8(d) prices EstimateReScalingOpt at five passes over 32 B per KeyLine; the kernel keeps a sequence's KeyLines in
        # registers / LDS across the passes, so what a launch has to move is ONE pass (a fraction of the roofline above 1 would
        # only say that the formula counts bytes nobody moves)
        "C.rescale": 32 * kn,
        "C.pose": 0,
    }
    return per_seq.get(group, 0) * nseq


def survey_bytes(group, kn, n_px, radius, nseq):
    """SURVEY.md 8(d)'s own per-unit figure where it differs from what a launch must move (`frac_survey_formula`): 84 B per
    KeyLine and EVALUATION, so 168 B for a two-chain launch; five passes over 32 B for EstimateReScalingOpt."""
    over = {"B.try_velrot2": 2 * 84 * kn, "C.rescale": 5 * 32 * kn, "B.build_field": 8 * n_px + 8 * 2 * radius * kn}
    return over[group] * nseq if group in over else algorithmic_bytes(group, kn, n_px, radius, nseq)


# kernel names (substrings of the rocprofv3 kernel name) behind each HIP-event group
GROUP_KERNELS = {
    "A.rgb_rowscan": ["k_rgb_rowscan"], "A.colscan": ["k_colscan"], "A.avg_rowscan": ["k_avg_rowscan"],
    "A.detect": ["k_detect"], "A.compact": ["k_strip_scan", "k_emit"], "A.join_retune": ["k_join_histo", "k_retune"],
    "A.level": ["k_level"], "A.fused": ["k_stage_a_fused"],
    "B.quantile": ["k_quantile"], "B.build_field": ["k_field_bin", "k_field_raster"], "B.tvr_prepare": ["k_tvr_prepare"],
    # (the float configuration's launches are k_try_velrot_f32 / k_try_velrot2_f32: a run holds one precision's kernels, never both)
    "B.try_velrot": ["k_try_velrot", "k_try_velrot_rw2", "k_try_velrot_f32"], "B.try_velrot2": ["k_try_velrot2", "k_try_velrot2_f32"], "B.lm_step": ["k_lm_step", "k_lm_step2"],
    "B.minimizer": ["k_try_velrot_lm"],
    "C.forward_match": ["k_fwd_key", "k_fwd_win", "k_fwd_apply"], "C.rotate": ["k_rot_from_state", "k_rotate", "k_fwd_apply_rotate"],
    "C.directed_matching": ["k_directed", "k_directed_fused"], "C.regularize_ekf": ["k_regularize", "k_ekf"], "C.rescale": ["k_rescale"],
}

PMC_FILE = os.path.join("profiles", "pmc_latest.json")


def fetch_calibration():
    """true bytes / FETCH_SIZE-reported bytes for the access patterns of this path, measured with known byte counts
    (tools/experiments/ubench_fetch.hip -> profiles/fetch_calibration.json): {"stream": f, "gather16": f, "gather2": f}.
    Falls back to the doubling MI355X_MICROARCH.md prescribes for wide streams."""
    path = os.path.join(ROOT, "profiles", "fetch_calibration.json")
    try:
        js = json.load(open(path))
        return {k: float(v) for k, v in js["factor"].items()}, "profiles/fetch_calibration.json"
    except (OSError, KeyError, ValueError):
        return {"stream": 2.0}, "MI355X_MICROARCH.md (gfx950: FETCH_SIZE x 2, calibrated for wide coalesced streams only)"


# Kernel groups whose reads are a mix of coalesced streams and small random gathers: bytes per KeyLine they read as
# streams.  FETCH_SIZE counts a request of a wide stream at half its bytes and a small random read at the 64 bytes it
# moves (profiles/fetch_calibration.json), so the counter R of such a kernel is stream/f_stream + gather/f_gather and the
# bytes it really moved are  stream + f_gather * (R - stream / f_stream)  with the stream bytes known exactly.
GROUP_STREAM_BYTES_PER_KL = {"B.try_velrot": 8 + 4 + 8 + 8 + 8 + 4 + 8,   # s_rho, m_num, p_m, rho, m_m, n_m, residual in (reweighted)
                             "B.try_velrot2": 8 + 4 + 8 + 8 + 8 + 4}       # the initialisation chains are not reweighted: no residual in


def pmc_kn(check_stamp=True):
    """KeyLines per frame of the run the committed counters were taken in (tools/gpu_round.sh stamps it), or None."""
    js = _pmc_file(check_stamp)
    return js.get("_kn") if js else None


def pmc_counters(group, nseq, check_stamp=True):
    """(FETCH_SIZE, WRITE_SIZE) in bytes per launch of `group`, raw, from the committed PMC passes; None if unavailable, taken at
    another batch size, or (check_stamp) taken with other sources than the running library's."""
    js = _pmc_file(check_stamp)
    if not js or js.get("_nseq") != nseq:
        return None
    ft = wt = 0.0
    found = False
    for sub in GROUP_KERNELS.get(group, []):
        f = w = n = 0.0
        for name, c in js.items():
            if not isinstance(c, dict) or not re.search(r"\b" + sub + r"\b", name):
                continue
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                k = c["FETCH_SIZE"]["calls"]
                f += c["FETCH_SIZE"]["mean"] * k
                w += c["WRITE_SIZE"]["mean"] * k
                n += k
        if n:
            ft += f / n * 1024.0
            wt += w / n * 1024.0
            found = True
    return (ft, wt) if found else None


def calibrated_traffic(group, nseq, kn, calib, stream_per_kl=None, check_stamp=True):
    """HBM bytes per launch of `group` from the committed counters and the calibration of profiles/fetch_calibration.json
    (see GROUP_STREAM_BYTES_PER_KL); (bytes, description of the formula) or (None, None)."""
    c = pmc_counters(group, nseq, check_stamp)
    if c is None:
        return None, None
    fetch, write = c
    fs, fg, fw = calib.get("stream", 2.0), calib.get("gather16", 1.0), calib.get("write", 1.0)
    if group in GROUP_STREAM_BYTES_PER_KL and "gather16" in calib:
        stream = (stream_per_kl or GROUP_STREAM_BYTES_PER_KL[group]) * kn * nseq
        gather = max(0.0, fetch - stream / fs) * fg
        return int(stream + gather + fw * write), (f"streams {stream / 1e6:.0f} MB (known) + {fg:.2f} x (FETCH_SIZE - streams / {fs:.2f}) "
                                                   f"= {gather / 1e6:.0f} MB of gathers + {fw:.2f} x WRITE_SIZE")
    return int(fs * fetch + fw * write), f"{fs:.2f} x FETCH_SIZE + {fw:.2f} x WRITE_SIZE (coalesced streams)"


def pmc_traffic(group, nseq, factor=2.0, check_stamp=True):
    """HBM bytes per launch of `group` from the committed rocprofv3 PMC passes (profiles/pmc_latest.json, made by
    tools/gpu_round.sh: separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of this same command).
    FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE is scaled by `factor` (2 = the gfx950 note in MI355X_MICROARCH.md).  None
    when the file is missing, was taken at another batch size, or lacks the kernel."""
    js = _pmc_file(check_stamp)
    if not js or js.get("_nseq") != nseq:
        return None
    total, found = 0.0, False
    for sub in GROUP_KERNELS.get(group, []):
        # template instantiations of one kernel (e.g. k_try_velrot<true,true>) are averaged by call count
        f = w = n = 0.0
        for name, c in js.items():
            if not isinstance(c, dict) or not re.search(r"\b" + sub + r"\b", name):
                continue
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                k = c["FETCH_SIZE"]["calls"]
                f += c["FETCH_SIZE"]["mean"] * k
                w += c["WRITE_SIZE"]["mean"] * k
                n += k
        if n:
            total += (factor * f / n + w / n) * 1024.0
            found = True
    return int(total) if found else None


def _usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def frame_stats(done_s, skip):
    """Per-frame intervals of a replay (seconds at which each frame was finished) after `skip` warm-up frames."""
    iv = np.diff(np.asarray(done_s)[skip - 1:]) if skip > 0 else np.diff(np.concatenate([[0.0], done_s]))
    return {"value": round(len(iv) / float(iv.sum()), 2), "unit": "frames/s", "ms_per_frame": round(float(iv.mean()) * 1e3, 3),
            "median_ms": round(float(np.median(iv)) * 1e3, 3), "p95_ms": round(float(np.percentile(iv, 95)) * 1e3, 3),
            "frames": int(len(iv))}


def pcie_inclusive(edgehip, params, frames, offs, nseq, steps, warmup, device):
    """The whole path with every frame crossing PCIe inside the timed region: page-locked host frames, DISTINCT per sequence
    (what B cameras or decoders would have filled), one asynchronous copy per step on the context's upload stream
    (edgehip_upload_rgb_pinned / edgehip_upload_grey8_pinned: the copy of step k+1 runs under the kernels of step k), then
    edgehip_process_frame.  RGB24 = 3 B per pixel, the reference's camera format; grey8 = the 1 B per pixel a mono camera /
    EuRoC delivers.  Four host buffers hold four consecutive pool frames of every sequence and are swept back and forth
    (continuous camera motion, same per-frame work as the resident-input line)."""
    P = 4
    out = {}
    for fmt in ("rgb24", "grey8"):
        try:
            eh = edgehip.EdgeHip(params, nseq=nseq, nslots=3, device=device)
            bufs = [eh.alloc_pinned_frames() if fmt == "rgb24" else eh.alloc_pinned_grey8() for _ in range(P)]
            for b, (arr, _) in enumerate(bufs):
                for s_ in range(nseq):
                    f = frames[tri(b + int(offs[s_]), len(frames))]
                    arr[s_] = f if fmt == "rgb24" else f[:, :, 1]
            up = eh.upload_rgb_pinned if fmt == "rgb24" else eh.upload_grey8_pinned

            def step(k):
                up(eh.next_slot(), bufs[tri(k, P)][1])
                eh.process_frame(FRAME_DT * k)
            for k in range(warmup):
                step(k)
            eh.sync()
            t0 = time.perf_counter()
            for k in range(warmup, warmup + steps):
                step(k)
            eh.sync()
            dt = time.perf_counter() - t0
            nav = eh.read_nav()
            bytes_step = nseq * eh.h * eh.w * (3 if fmt == "rgb24" else 1)
            out[fmt] = {"value": round(nseq * steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 4),
                        "host_to_device_GBs": round(bytes_step * steps / dt / 1e9, 2), "MB_per_frame": round(bytes_step / nseq / 1e6, 3),
                        "steps": steps, "warmup": warmup, "sequences": nseq,
                        "estimation_ok": f"{int(sum(n.estimation_ok for n in nav))}/{nseq}"}
            eh.sync()
            for _, ptr in bufs:
                eh.free_pinned(ptr)
            eh.close()
        except Exception as e:
            out[fmt] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
    return out


def _cpu_imu_worker(job):
    """The reference's ImuMode > 0 frame order (oracle/ref_harness.cpp::ref_process_frame_imu = rebvo_second_t.cpp:182-336,
    519-606 over the reference's own tracker, ExtRotVel, BiasCorrect and ScaleEstimator) on one sequence, in a process of its
    own (ScaleEstimator keeps its histories in function-local statics).  Returns per-frame seconds and the trajectory."""
    w, h, pool_frames, idx, imu_rows, dt_f, imu_over = job
    import ctypes as C
    from oracle import oracle
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
    L = orc.lib
    L.ref_imu_setup.argtypes = [C.c_void_p, C.POINTER(oracle.ImuParams)]
    L.ref_imu_setup.restype = None
    L.ref_process_frame_imu.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(oracle.ImuIntegrated), C.POINTER(oracle.NavImu)]
    L.ref_process_frame_imu.restype = C.c_int
    ip = oracle.euroc_imu_params(**imu_over)
    L.ref_imu_setup(orc.ctx, C.byref(ip))
    secs, pos, pose, vel, rotlie, ok = [], [], [], [], [], []
    for k, i in enumerate(idx):
        nav = oracle.NavImu()
        f = np.ascontiguousarray(pool_frames[i], np.uint8)
        rec = oracle.ImuIntegrated.from_row(imu_rows[k])
        t1 = time.perf_counter()
        L.ref_process_frame_imu(orc.ctx, f.ctypes.data, dt_f * k, C.byref(rec), C.byref(nav))
        secs.append(time.perf_counter() - t1)
        pos.append(np.array(nav.Pos[:])); pose.append(np.array(nav.Pose[:])); vel.append(np.array(nav.Vel[:]))
        rotlie.append(np.array(nav.RotLie[:])); ok.append(int(nav.estimation_ok))
    orc.close()
    return np.array(secs), np.array(pos), np.array(pose), np.array(vel), np.array(rotlie), np.array(ok)


def _cpu_worker(job):
    """One CPU-reference sequence in its own process (node-saturating mode): the reference's own two-thread overlap on
    `nfr` frames; returns the seconds from the first timed frame to the last."""
    kind, nfr, npool, seed, threads, w, h = job
    from oracle import oracle
    from rebvo_amd import synth
    frames = np.stack([f for f, _, _ in synth.billboard_sequence(w, h, min(npool, 8), seed=seed)])
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
    idx = [tri(k, len(frames)) for k in range(8 + nfr)]
    done, _ = orc.run_sequence(frames, idx, threads=threads)
    return float(done[-1] - done[7])


_PARITY = {}


def _parity_init(frames_arr, cfg):
    """Pool initialiser of wide_parity: the frame pool and the parameter set, once per worker process."""
    _PARITY["frames"], _PARITY["cfg"] = frames_arr, cfg


def _parity_worker(job):
    """The CPU reference on one sequence of a batch (frame indices `idx` into the shared pool), from frame 0: per-frame V, W,
    Pos, Pose, the counts, and the frames whose result the reference's own arithmetic leaves undecided (a KeyLine detected
    exactly on a half pixel whose re-projection at X = 0 rounds either way: oracle.half_pixel_keylines)."""
    s_, idx = job
    from oracle import oracle
    kind, w, h, dt = _PARITY["cfg"][:4]
    op = oracle.tum_params(w, h, use_undistort=1) if kind == "tum" else oracle.euroc_params(w, h)
    orc = oracle.Oracle("ref", op)
    if len(_PARITY["cfg"]) > 4 and _PARITY["cfg"][4]:
        orc.set_tracker_f32(1)   # the reference's float instantiation of the tracker (--tracker-f32)
    fr = _PARITY["frames"]
    V, Wv, Pos, Pose, cnt, knife = [], [], [], [], [], []
    for k, i in enumerate(idx):
        old = orc.keylines(orc.cur_slot()).copy() if k else None
        _, nav = orc.process_frame(fr[i], dt * k)
        V.append(nav.V[:]); Wv.append(nav.W[:]); Pos.append(nav.Pos[:]); Pose.append(nav.Pose[:])
        cnt.append((nav.kn, nav.klm_num, nav.estimation_ok))
        if k and oracle.half_pixel_keylines(old, orc.field(orc.cur_slot())[:, :, 1], op.ppx, op.ppy, nav.s_rho_q, op.w, op.h):
            knife.append(k)
    orc.close()
    return s_, np.array(V), np.array(Wv), np.array(Pos), np.array(Pose).reshape(-1, 3, 3), np.array(cnt), knife


def wide_parity(log, seqs, idx_of, frames_arr, cfg, first_timed, procs, tol_rel=1e-6, tol_abs=1e-9):
    """Free-running parity of MANY sequences of a batch against the CPU reference replaying the same frames from frame 0, in a
    pool of `procs` reference processes.  log: the device's nav records of frames 0..n-1 ([n, nseq] structured array);
    idx_of(s) -> the n pool indices of sequence s.  Per frame the tests' bound: |dV|, |dW| <= 1e-6 * step + 1e-9 (and the same
    KeyLine count).  Returns (summary, {s: (Pos, Pose, V, W) reference trajectory over the timed frames}).
    A sequence that leaves the reference does so for good (discrete match decisions differ from then on), so what is
    reported is how many are outside tolerance at the last frame, where each of those first left, and whether that frame is
    one the reference itself leaves undecided (a knife-edge frame: two runs of the reference — another LAPACK, another CPU
    model — part on the same frames, profiles/r03_knife_edge_*)."""
    import multiprocessing as mp
    n = log.shape[0]
    jobs = [(int(s_), [int(i) for i in idx_of(s_)]) for s_ in seqs]
    with mp.get_context("spawn").Pool(max(1, min(procs, len(jobs))), initializer=_parity_init, initargs=(frames_arr, cfg)) as pool_:
        res = pool_.map(_parity_worker, jobs)
    trajs, departed, frames_out = {}, [], 0
    max_dv_inside = 0.0
    for s_, V, Wv, Pos, Pose, cnt, knife in res:
        trajs[s_] = {k - first_timed: (Pos[k], Pose[k], V[k], Wv[k]) for k in range(first_timed, n)}
        first = None
        path = 0.0
        for k in range(1, n):
            if not (np.all(np.isfinite(V[k])) and np.all(np.isfinite(Wv[k]))):
                continue
            tol = tol_rel * (np.linalg.norm(V[k]) + np.linalg.norm(Wv[k])) + tol_abs
            d = max(np.max(np.abs(log[k, s_]["V"] - V[k])), np.max(np.abs(log[k, s_]["W"] - Wv[k])))
            bad = d > tol or int(log[k, s_]["kn"]) != int(cnt[k][0])
            if bad:
                frames_out += 1
                if first is None:
                    first = k
            elif first is None:
                max_dv_inside = max(max_dv_inside, float(d))
            path += float(np.linalg.norm(V[k]))
        last = n - 1
        tol_l = tol_rel * (np.linalg.norm(V[last]) + np.linalg.norm(Wv[last])) + tol_abs
        out_last = bool(max(np.max(np.abs(log[last, s_]["V"] - V[last])), np.max(np.abs(log[last, s_]["W"] - Wv[last]))) > tol_l or
                        np.max(np.abs(log[last, s_]["Pos"] - Pos[last])) > tol_rel * path + tol_abs)
        if first is not None or out_last:
            departed.append({"sequence": s_, "first_frame_outside_tolerance": first, "knife_edge_frame": bool(first in knife) if first is not None else None,
                             "outside_tolerance_at_last_frame": out_last,
                             "position_error_at_last_frame": float(np.linalg.norm(log[last, s_]["Pos"] - Pos[last])),
                             "knife_edge_frames_of_the_reference": knife[:12]})
    summary = {"sequences_checked": len(res), "frames_per_sequence": n, "reference_processes": min(procs, len(jobs)),
               "sequences_outside_tolerance_at_last_frame": int(sum(d_["outside_tolerance_at_last_frame"] for d_ in departed)),
               "departures": departed,
               "departures_on_knife_edge_frames": int(sum(1 for d_ in departed if d_["knife_edge_frame"])),
               "departures_elsewhere": int(sum(1 for d_ in departed if d_["knife_edge_frame"] is False)),
               "max_abs_dVW_while_inside_tolerance": max_dv_inside,
               "tolerance": f"per frame |dV|, |dW| <= {tol_rel:g} * (|V| + |W|) + {tol_abs:g} and the same KeyLine count; last frame also |dPos| <= {tol_rel:g} * path + {tol_abs:g}"}
    return summary, trajs


def pose_rmse(gpu_trajs, cpu_trajs, kind):
    """BASELINE.json's "pose RMSE vs CPU ref": trajectories of a few sequences of the batch over the timed frames, HIP
    path against the CPU reference run on the same frame order from the same start (frame 0).  Position in the
    (up-to-scale) map units of NavData::Pos, rotation as the angle of Pose_gpu * Pose_cpu^T, V/W = the per-frame tracker
    outputs.  RMS over all checked frames of all checked sequences."""
    dp, dr, dv, dw, path, nfr = [], [], [], [], 0.0, 0
    per_seq = {}
    for s, cpu_traj in cpu_trajs.items():
        n0 = len(dp)
        gpu_traj = gpu_trajs.get(s) if gpu_trajs else None
        ks = sorted(k for k in cpu_traj if gpu_traj and k < len(gpu_traj))
        for i, k in enumerate(ks):
            gp, gR, gv, gw = gpu_traj[k]
            cp, cR, cv, cw = cpu_traj[k]
            dp.append(float(np.sum((gp - cp) ** 2)))
            # angle of gR * cR^T: ||gR - cR||_F = 2*sqrt(2)*sin(angle/2) (no arccos of a number next to 1)
            dr.append(float((2.0 * np.arcsin(min(1.0, np.linalg.norm(gR - cR) / (2.0 * np.sqrt(2.0))))) ** 2))
            dv.append(float(np.sum((gv - cv) ** 2)))
            dw.append(float(np.sum((gw - cw) ** 2)))
            if i:
                path += float(np.linalg.norm(cp - cpu_traj[ks[i - 1]][0]))
        nfr += len(ks)
        if len(dp) > n0:
            per_seq[int(s)] = float(np.sqrt(np.mean(dp[n0:])))
    if not nfr:
        return None
    rms = lambda a: float(np.sqrt(np.mean(a)))
    return {"position": rms(dp), "rotation_rad": rms(dr), "V": rms(dv), "W": rms(dw), "frames": nfr,
            "sequences": sorted(cpu_trajs), "position_per_sequence": per_seq, "path_length": path,
            "position_rel": rms(dp) / path * len(cpu_trajs) if path > 0 else None,
            "vs": "CPU " + kind + " on the same frames, both started at frame 0 (tests bound |dV|,|dW| by 1e-6 relative)"}


def _traj_of_log(log, s):
    return [(log[k, s]["Pos"].copy(), log[k, s]["Pose"].reshape(3, 3).copy(), log[k, s]["V"].copy(), log[k, s]["W"].copy())
            for k in range(log.shape[0])]


def _cpu_traj(oracle, params, frames_of_step, first, count, kind=None):
    """The CPU reference on frames_of_step(0..first+count-1); its trajectory over the last `count` frames."""
    orc = oracle.Oracle(kind or ("ref" if oracle.available("ref") else "port"), params)
    out = {}
    for k in range(first + count):
        _, nav = orc.process_frame(frames_of_step(k), FRAME_DT * k)
        if k >= first:
            out[k - first] = (np.array(nav.Pos[:]), np.array(nav.Pose[:]).reshape(3, 3), np.array(nav.V[:]), np.array(nav.W[:]))
    orc.close()
    return out


class Replay:
    """`nseq` sequences in lock-step out of an HBM-resident frame pool: frame index of sequence s at step k =
    index_of(k)[s]."""

    def __init__(self, edgehip, params, nseq, pool_t, pool_frames, index_of, device, contexts=1, imu_params=None, imu_of=None):
        self.C = max(1, contexts)
        self.B = nseq // self.C
        self.ehs = [edgehip.EdgeHip(params, nseq=self.B, nslots=3, device=device) for _ in range(self.C)]
        self.pool_t, self.pool_frames, self.index_of = pool_t, pool_frames, index_of
        self.imu_of = imu_of          # k -> integrated IMU records of all sequences for the interval that ends with frame k
        if imu_params is not None:
            for e in self.ehs:
                e.imu_enable(imu_params)

    def step(self, k):
        idx = np.ascontiguousarray(self.index_of(k), dtype=np.int32)
        imu = self.imu_of(k) if self.imu_of else None
        for ci, e in enumerate(self.ehs):
            e.bind_rgb_indexed(e.next_slot(), self.pool_t.data_ptr(), self.pool_frames, idx[ci * self.B:(ci + 1) * self.B])
            if imu is not None:
                e.set_imu(imu[ci * self.B:(ci + 1) * self.B])
            e.process_frame(FRAME_DT * k)

    def sync(self):
        for e in self.ehs:
            e.sync()

    def close(self):
        for e in self.ehs:
            e.close()


def timed_replay(rp, K, Wm, profile=False):
    """Wm warm-up steps, then K steps between synchronisations; returns (seconds, breakdown in us per step or None)."""
    eh = rp.ehs[0]
    breakdown = None
    if profile:
        prof_steps = min(4, max(1, Wm // 4))
        for k in range(Wm - prof_steps):
            rp.step(k)
        rp.sync()
        eh.profile_enable(True)
        eh.profile_select(None)
        for k in range(Wm - prof_steps, Wm):
            rp.step(k)
        prof = eh.profile_read()
        eh.profile_enable(False)
        breakdown = {g: (ms / prof_steps * 1e3, calls // prof_steps) for g, (ms, calls) in prof.items() if calls}
    else:
        for k in range(Wm):
            rp.step(k)
    rp.sync()
    t0 = time.perf_counter()
    for k in range(Wm, Wm + K):
        rp.step(k)
    rp.sync()
    return time.perf_counter() - t0, breakdown


def _write_surface_imu_csv(path, pool, n_intervals, t0, dt, rate_hz=200.0):
    """IMU samples (EuRoC csv: t, gyro xyz, accel xyz; seconds) for a camera that walks the `pool` rendered frames back and forth
    (frame of time index m = tri(m, pool)): the body rate of interval m-1 -> m from the known rotation between the two frames, a
    constant gyro bias on top, gravity in the camera frame of the frame the interval ends at (as bench.py --imu synthesises its
    integrated records)."""
    from rebvo_amd import synth
    tw = synth.smooth_trajectory(pool, 13)
    poses = [R for _, R, _ in synth.billboard_sequence(8, 8, pool, seed=11)]
    bias = np.array([0.004, -0.002, 0.003])
    with open(path, "w") as f:
        f.write("#timestamp [s],w_x,w_y,w_z,a_x,a_y,a_z\n")
        ts, step = t0 - 12.0 / rate_hz, 1.0 / rate_hz
        while ts < t0 + dt * n_intervals:
            m = max(int(np.floor((ts - t0) / dt)) + 1, 1)          # the interval (m-1, m] this sample lies in
            a, b = tri(m - 1, pool), tri(m, pool)
            rot = tw[a, 3:] if b > a else (-tw[b, 3:] if b < a else np.zeros(3))
            gyro = -rot / dt + bias
            acc = -(poses[b] @ np.array([0.0, 9.8, 0.0]))
            f.write("%.9f,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n" % (ts, *gyro, *acc))
            ts += step


def host_surface(params, frames, w, h):
    """Frames per second THROUGH the plugin surface (requestCustomCamBuffer -> copyFrom -> releaseCustomCamBuffer, results
    through getNav; include/rebvo/rebvo.h:548-609 of the reference): rebvo_amd/lib/surface_replay drives 1, 8 and 64
    rebvo::REBVO objects, the 8 and the 64 as ONE batch group each (&GPU BatchGroup: one shared context, lock-step, page-locked
    camera rings, asynchronous uploads under the frames before; rebvo_amd/host/src/batch_group.cpp).  Every frame crosses PCIe
    inside the timed region, so these are PCIe-inclusive figures by construction: the application writes RGB24 (1.08 MB per frame,
    one copyFrom each, up to 16 producer threads); the synthetic frames are mono (R = G = B), so with &GPU MonoUpload at its default
    the group sends their 8-bit planes (0.36 MB; `objects_64_rgb24_fps` is the same run with MonoUpload = 0: 1.08 MB per frame).  No callback is registered
    for the three headline numbers (KeyLines stay in HBM); `objects_8_with_callbacks_fps` adds one per object (AoS KeyLines back
    to the host for every frame; two steps in flight since round 6: edgehip_export_keylines), `objects_8_imu_fps` is eight ImuMode = 2
    members in one group (the device-side IMU branch behind the surface), `objects_8_stereo_fps` eight StereoAvaiable members in one group (a pair
    frame per main frame; the pair images are the main images: the pair path's cost, not a depth result), `objects_256_fps` / `objects_1024_fps` the same leg with 256 /
    1024 cameras and 16 producer threads (the better of two runs each — and so for the 64-object legs —, both in `detail.*.fps_of_each_run`; `detail` carries where the producers' time goes).  Run lengths: 600 / 400 / 240 / 300 frames per object, the first 60 / 50 / 40 / 40 untimed — the
    application runs up to three frames ahead of the tracker (the camera ring) and the last frames drain, which weighed 15-30 % in
    the 36-frame runs of the first version of this leg."""
    import subprocess
    import tempfile
    from rebvo_amd import config
    exe = os.path.join(ROOT, "rebvo_amd", "lib", "surface_replay")
    if not os.path.exists(exe):
        return {"error": "rebvo_amd/lib/surface_replay not built"}
    out = {"what": "frames/s through requestCustomCamBuffer/getNav, PCIe inside (mono frames as 8-bit planes; *_rgb24: as RGB24); 8 and 64 objects = one batch group each"}
    with tempfile.TemporaryDirectory() as td:
        cfg, raw = os.path.join(td, "cfg"), os.path.join(td, "frames.rgb24")
        config.write_global_config(cfg, params)
        np.stack(frames).tofile(raw)
        cfg_mono = cfg
        for name, n, k, wm, extra in (("single_camera_fps", 1, 600, 60, []), ("objects_8_fps", 8, 400, 50, ["--group", "g8"]),
                                      ("objects_64_fps", 64, 240, 40, ["--group", "g64"]),
                                      ("objects_8_with_callbacks_fps", 8, 300, 40, ["--group", "g8cb", "--callback"]),
                                      ("single_camera_with_callback_fps", 1, 600, 60, ["--callback"]),
                                      ("objects_64_rgb24_fps", 64, 240, 40, ["--group", "g64c", "--rgb24"]),
                                      ("objects_8_imu_fps", 8, 300, 40, ["--group", "g8imu", "--imu"]),
                                      ("objects_8_stereo_fps", 8, 300, 40, ["--group", "g8st", "--stereo"]),
                                      ("objects_256_fps", 256, 120, 30, ["--group", "g256"]),
                                      ("objects_1024_fps", 1024, 60, 15, ["--group", "g1024"])):
            try:
                cfg = cfg_mono
                if "--rgb24" in extra:
                    cfg = os.path.join(td, "cfg_rgb24")
                    config.write_global_config(cfg, params, gpu=dict(mono=0))
                    extra = [e for e in extra if e != "--rgb24"]
                if "--imu" in extra:
                    # ImuMode = 2 members (what GlobalConfig_EuRoC ships with) in one group: one IMU file on the common time line, object i
                    # enters it i frames late (--stagger), the gyro / accelerometer samples synthesised from the known camera motion
                    cfg = os.path.join(td, "cfg_imu")
                    imu_csv = os.path.join(td, "imu.csv")
                    _write_surface_imu_csv(imu_csv, len(frames), k + n + 2, 1.0, FRAME_DT)
                    config.write_global_config(cfg, params, imu=dict(mode=2, file=imu_csv, time_scale=1.0, InitBiasFrameNum=3))
                    extra = [e for e in extra if e != "--imu"] + ["--stagger"]
                if "--stereo" in extra:
                    # StereoAvaiable members in one group (round 6): a pair frame per main frame through requestStereoCustomCamBuffer, the rig
                    # inside the group's edgehip_process_frame.  The pair images ARE the main images here (the same file): this leg times the
                    # pair path (second ring, second copy, stage A of the pair slot, stereo matching, fusion), it is not a depth result
                    cfg = os.path.join(td, "cfg_stereo")
                    config.write_global_config(cfg, params, dataset=("unused/", "unused.csv", 1.0),
                                               stereo=dict(dir="unused/", file="unused.csv", ppx=params.ppx, ppy=params.ppy, zfx=params.zfx, zfy=params.zfy))
                    extra = [e for e in extra if e != "--stereo"] + ["--stereo", raw]
                # the 256- and 1024-camera legs are bound by the application's 16 producer threads (copyFrom + the mono test: ~2.5 MB of host
                # memory traffic per frame) on a host this process shares with other tenants: two runs, the better one reported, both kept
                runs = []
                for _ in range(2 if n >= 64 else 1):   # (64 objects too: 39.6 … 82 k frames/s over the round's runs with the same binaries, a leg bound by the producers' side)
                    t_leg = time.perf_counter()
                    r = subprocess.run([exe, cfg, raw, str(len(frames)), str(n), str(k), "1", str(FRAME_DT), "--warmup", str(wm),
                                        "--threads", str(min(16, n))] + extra, capture_output=True, text=True, timeout=120)
                    js = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
                    if js is None:
                        raise RuntimeError(f"rc {r.returncode}: {(r.stdout + r.stderr)[-200:]}")
                    js["process_wall_s"] = round(time.perf_counter() - t_leg, 2)   # start-up, the run, shutdown of the whole process
                    runs.append(js)
                js = max(runs, key=lambda j_: j_["fps"])
                if len(runs) > 1:
                    js["fps_of_each_run"] = [j_["fps"] for j_ in runs]
                out[name] = js["fps"]
                out.setdefault("detail", {})[name] = js
            except Exception as e:
                out[name] = None
                out.setdefault("errors", {})[name] = f"{type(e).__name__}: {e}"[:200]
    if out.get("single_camera_fps"):
        out["single_camera_ms_per_frame"] = round(1e3 / out["single_camera_fps"], 4)
    return out


LINE_LIMIT = 4096      # bytes of the one JSON line (the driver reads it out of a bounded stdout tail: round 4's 22 KB line did not parse)
EXTRAS_FILE = "bench_extras.json"


def _r(x, nd=4):
    """Numbers of the one line: rounded to what they are known to."""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    return x


def compact_line(full):
    """The ONE line on stdout, <= LINE_LIMIT bytes: the contract's keys, `config` (the workload named), `roofline` of the dominant
    kernel, `cpu_baseline` with the three modes as three numbers, `pose_rmse` with the parity counts, and the single-camera and
    plugin-surface figures.  Everything else — per-kernel rooflines, per-sequence maps, the sweep, notes, the `--extras` legs — is
    in bench_extras.json next to this file (and on stderr)."""
    c = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    wl = str(c.get("workload") or "")
    line["config"] = {"workload": wl if len(wl) <= 300 else wl[:297] + "...",
                      "input": (str(c["input"])[:160] if c.get("input") else None)}
    for k in ("sequences_per_gpu", "frames_per_step", "keylines_per_frame", "keylines_per_frame_timed_mean", "tryvelrot_evals_per_frame",
              "estimation_ok", "nav_gather", "algorithmic_MB_per_frame", "whole_path_hbm_frac", "frames_per_s", "tracker"):
        if c.get(k) is not None:
            line["config"][k] = c[k]
    if isinstance(c.get("nav_gather_info"), dict):
        line["config"]["nav_gather_equals_device_log"] = c["nav_gather_info"].get("equals_device_log")
    ds = c.get("dataset")
    line["config"]["dataset"] = ({"list": str(ds.get("list"))[-80:], "frames_in_pool": ds.get("frames_in_pool"),
                                  "mean_frame_interval_s": ds.get("mean_frame_interval_s")} if isinstance(ds, dict) else None)
    rf = full.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = {k: _r(rf.get(k), 5) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_on_traffic",
                                                         "frac_survey_formula", "traffic", "launch_us", "algorithmic_bytes_per_launch",
                                                         "launches_timed", "issue_frac", "traffic_live", "issue_live") if k in rf}
        if rf.get("traffic") is None and rf.get("traffic_note"):
            line["roofline"]["traffic_note"] = str(rf["traffic_note"])[:120]
    else:
        line["roofline"] = None
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        o = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "ms_per_frame", "median_ms", "p95_ms", "usable_cores", "cpu_model") if k in cb}
        if cb.get("sample"):
            o["sample"] = str(cb["sample"])[:200]
        if cb.get("error"):
            o["error"] = str(cb["error"])[:120]
        if isinstance(cb.get("modes"), dict):
            o["modes"] = {k: (v.get("value") if isinstance(v, dict) else v) for k, v in cb["modes"].items()}
            ns = cb["modes"].get("node_saturating")
            if isinstance(ns, dict) and ns.get("cores"):
                o["node_saturating_cores"] = ns["cores"]
        line["cpu_baseline"] = o
    else:
        line["cpu_baseline"] = None
    pr = full.get("pose_rmse")
    if isinstance(pr, dict):
        o = {k: _r(pr.get(k)) for k in ("position", "rotation_rad", "V", "W", "position_rel", "frames", "ranks", "error") if pr.get(k) is not None}
        par = pr.get("free_running_parity") or {}
        seqs = pr.get("sequences")
        o["sequences_checked"] = par.get("sequences_checked", pr.get("sequences_checked", len(seqs) if seqs else None))
        o["outside_tolerance"] = par.get("sequences_outside_tolerance_at_last_frame")
        o["departures_on_knife_edge_frames"] = par.get("departures_on_knife_edge_frames")
        o["departures_elsewhere"] = par.get("departures_elsewhere")
        if par.get("max_abs_dVW_while_inside_tolerance") is not None:
            o["max_abs_dVW_inside_tolerance"] = _r(par["max_abs_dVW_while_inside_tolerance"])
        o["tolerance"] = str(par.get("tolerance") or "per frame |dV|,|dW| <= 1e-6*(|V|+|W|)+1e-9 and equal KeyLine count")[:110]
        tfz = pr.get("teacher_forced")
        if isinstance(tfz, dict):
            o["teacher_forced"] = {k: _r(tfz.get(k)) for k in ("sequences", "frames", "frames_outside_tolerance", "max_rel_dX", "tolerance") if k in tfz}
        line["pose_rmse"] = o
    else:
        line["pose_rmse"] = None
    if isinstance(c.get("nav_gather_info"), dict):
        for k in ("ranks", "records", "device_to_wire"):
            if c["nav_gather_info"].get(k) is not None:
                line["config"]["nav_gather_" + k] = c["nav_gather_info"][k]
    for k in ("stage_a_hbm_frac", "single_sequence_ms_per_frame", "scaling_measured", "invalid_as_measurement", "launched_by"):
        if full.get(k) is not None:
            line[k] = full[k]
    hs = full.get("host_surface")
    if isinstance(hs, dict):
        line["host_surface"] = {k: v for k, v in hs.items() if isinstance(v, (int, float, str, bool)) and len(str(v)) <= 80}
    kus = full.get("kernel_us_per_step")
    if isinstance(kus, dict):   # the five heaviest groups, us per step
        line["kernel_us_per_step_top"] = dict(sorted(kus.items(), key=lambda kv: -kv[1])[:5])
    line["extras_file"] = EXTRAS_FILE
    # the bound holds whatever the run produced: drop the optional objects, longest first, until the line fits
    for k in ("kernel_us_per_step_top", "host_surface", "extras_file"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(k, None)
    if len(json.dumps(line)) > LINE_LIMIT:
        line["config"] = {"workload": wl[:120]}
    return line


def emit(full):
    """bench_extras.json + stderr get everything; stdout gets compact_line(full), one line."""
    text = json.dumps(full, indent=1, default=str)
    try:
        with open(os.environ.get("BENCH_EXTRAS_FILE") or os.path.join(ROOT, EXTRAS_FILE), "w") as f:
            f.write(text + "\n")
    except OSError as e:
        print(f"bench.py: {EXTRAS_FILE} not written: {e}", file=sys.stderr)
    print(text, file=sys.stderr)
    out = json.dumps(compact_line(full), allow_nan=False, default=str)
    print(out)
    sys.stdout.flush()


def library_source_sha():
    """First 16 hex digits of the sha256 over the library's sources (rebvo_amd/csrc/*.hip, *.h, include/edgehip.h, in name order):
    the identity of the code the committed counters were taken with (a rebuilt .so of the same sources keeps it)."""
    import glob
    import hashlib
    hsh = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "rebvo_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "rebvo_amd", "csrc", "*.h")))
    for fn in files + [os.path.join(ROOT, "include", "edgehip.h")]:
        try:
            hsh.update(os.path.basename(fn).encode() + b"\0" + open(fn, "rb").read())
        except OSError:
            return None
    return hsh.hexdigest()[:16]


_PMC_CACHE = {}


def _pmc_file(check_stamp=True):
    """profiles/pmc_latest.json if it exists and (check_stamp) was taken with the sources of the library that is running."""
    key = bool(check_stamp)
    if key not in _PMC_CACHE:
        js = None
        try:
            js = json.load(open(os.path.join(ROOT, PMC_FILE)))
            if check_stamp and (not js.get("_src_sha") or js.get("_src_sha") != library_source_sha()):
                js = None
        except (OSError, ValueError):
            js = None
        _PMC_CACHE[key] = js
    return _PMC_CACHE[key]


LIVE_PMC = {"used": False, "note": "not attempted", "issue": None}


def live_pmc_passes(child_argv, nseq, budget_s=90):
    """The counters of THIS run, observed by the command itself: more processes of this file under `rocprofv3 --kernel-trace --pmc ...`
    (no trace domain besides the kernel trace), same batch, steps and warm-up as the parent, CPU legs and extras off:
      FETCH_SIZE, WRITE_SIZE  (separate passes: the two do not share one on gfx950) -> their per-kernel means replace
                              profiles/pmc_latest.json for the rest of this process (same format, same calibration), so
                              `roofline.traffic` is what the driver's own box moved, not a committed constant;
      SQ_ACTIVE_INST_VALU + GRBM_GUI_ACTIVE (one pass, optional) -> `issue_frac` per kernel group (tools/sq_summary.py's definition,
                              weighted by cycles) instead of profiles/sq_latest.json.
    Any failure of the first two (no rocprofv3, a pass that times out or leaves no csv) keeps the committed counters under their
    source stamp; LIVE_PMC says which it was."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        LIVE_PMC["note"] = "rocprofv3 not on PATH"
        return False
    js = {}
    t_all = time.time()
    kn = None
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir=os.environ.get("TMPDIR") or "/tmp")

    def one_pass(tag, counters):
        nonlocal kn
        out = os.path.join(tmp, tag)
        cmd = [exe, "--kernel-trace", "--pmc"] + list(counters) + ["--output-format", "csv", "-d", out, "-o", "pmc", "--",
                                                                  sys.executable, os.path.abspath(__file__)] + child_argv
        env = dict(os.environ, BENCH_LIVE_PMC_CHILD="1", BENCH_EXTRAS_FILE=os.path.join(tmp, tag + "_extras.json"), TMPDIR=tmp)
        left = budget_s - (time.time() - t_all)
        if left < 20:
            return f"no time left for the {tag} pass"
        # (a process group of its own: a pass that overruns is ended as a whole — rocprofv3 AND the bench process under it)
        proc = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            p_out, p_err = proc.communicate(timeout=left)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                proc.kill()
            proc.communicate()
            return f"the {tag} pass did not finish in {left:.0f} s"
        if proc.returncode != 0:
            return f"the {tag} pass exited {proc.returncode}: " + (p_err or "")[-160:].replace("\n", " ")
        try:
            line = p_out[p_out.rindex('{"metric"'):]
            kn = json.loads(line)["config"].get("keylines_per_frame_timed_mean") or kn
        except (ValueError, KeyError):
            pass
        acc = {}
        for fn in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                if row.get("Counter_Name") not in counters:
                    continue
                a_ = acc.setdefault((row["Kernel_Name"], row["Counter_Name"]), [0, 0.0])
                a_[0] += 1
                a_[1] += float(row["Counter_Value"])
        if not acc:
            return f"the {tag} pass left no counter_collection.csv rows"
        for (k, c), (n, tot) in acc.items():
            js.setdefault(k, {})[c] = {"calls": n, "mean": tot / n}
        return None
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            err = one_pass(counter, (counter,))
            if err:
                LIVE_PMC["note"] = err
                return False
        err = one_pass("SQ", ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"))
        if err:
            LIVE_PMC["issue_note"] = err
        else:
            issue = {}
            for g, subs in GROUP_KERNELS.items():
                busy = cyc = 0.0
                for k, c in js.items():
                    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c and any(re.search(r"\b" + s_ + r"\b", k) for s_ in subs):
                        n = c["GRBM_GUI_ACTIVE"]["calls"]
                        busy += n * 4.0 * c["SQ_ACTIVE_INST_VALU"]["mean"] / 1024.0
                        cyc += n * c["GRBM_GUI_ACTIVE"]["mean"] / 8.0
                if cyc > 0:
                    issue[g] = round(busy / cyc, 4)
            LIVE_PMC["issue"] = issue or None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    js["_nseq"] = nseq
    js["_kn"] = kn
    js["_src_sha"] = library_source_sha()
    js["_command"] = "live: python bench.py " + " ".join(child_argv)
    _PMC_CACHE[True] = _PMC_CACHE[False] = js
    LIVE_PMC.update(used=True, seconds=round(time.time() - t_all, 1), kernels=len([k for k in js if not k.startswith("_")]),
                    note="FETCH_SIZE and WRITE_SIZE passes" + (" and an SQ_ACTIVE_INST_VALU + GRBM_GUI_ACTIVE pass" if LIVE_PMC["issue"] else "") +
                         " of this command, run by this process on this box")
    return True


def pmc_stamp_note():
    try:
        js = json.load(open(os.path.join(ROOT, PMC_FILE)))
    except (OSError, ValueError):
        return "no committed counters"
    have, want = js.get("_src_sha"), library_source_sha()
    return f"counters stamped {have}, library sources {want}: " + ("match" if have and have == want else "NO match, traffic = null")


def issue_fracs():
    """{group: share of the kernel's busy cycles in which VALU instructions issue} from this run's own SQ pass (live_pmc_passes) or from
    the committed SQ-counter passes (profiles/sq_latest.json, tools/gpu_round.sh), under the same source stamp as the HBM counters; {} otherwise."""
    if LIVE_PMC.get("issue"):
        return dict(LIVE_PMC["issue"])
    try:
        js = json.load(open(os.path.join(ROOT, "profiles", "sq_latest.json")))
        if not js.get("_src_sha") or js.get("_src_sha") != library_source_sha():
            return {}
        return {g: v for g, v in js.get("issue_frac", {}).items()}
    except (OSError, ValueError):
        return {}


def other_configs(args):
    """--extras: the other BASELINE configurations and the ImuMode=2 line, each measured by this same file in a process of its own
    (short runs in the driver's form); each entry is that run's own record, condensed."""
    import subprocess
    extras = {}
    cf = str(min(args.cpu_frames, 60))

    def run(flags, timeout):
        out = subprocess.run([sys.executable, os.path.abspath(__file__)] + flags, capture_output=True, text=True, timeout=timeout)
        for ln in reversed(out.stdout.splitlines()):
            if ln.startswith("{"):
                return json.loads(ln)
        raise RuntimeError(f"no JSON line (rc {out.returncode}): {out.stderr[-200:]}")
    for name, flags in (("stage_a", ["--config", "stage_a", "--cpu-frames", cf]),
                        ("tum_undistort", ["--config", "tum_undistort", "--cpu-frames", cf, "--cpu-procs", "0"]),
                        ("imu", ["--imu", "--cpu-frames", cf, "--cpu-procs", "0"]),
                        ("tracker_f32", ["--tracker-f32", "--cpu-frames", cf, "--cpu-procs", "0"])):
        try:
            t_sub = time.perf_counter()
            js = run(["--no-extras", "--steps", "20", "--warmup", "5", "--nseq", str(args.nseq)] + flags, 420)
            extras[name] = js
            extras[name]["command"] = "python bench.py --no-extras --steps 20 --warmup 5 " + " ".join(flags)
            extras[name]["wall_s"] = round(time.perf_counter() - t_sub, 1)
        except Exception as e:
            extras[name] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
    # ... and what GlobalConfig_EuRoC ships with for ONE camera (ImuMode=2, one sequence per launch; best of three: a single
    # camera's rate moves with the host's launch speed and with the clocks a lightly loaded GPU settles at)
    try:
        best = None
        for _ in range(3):
            js = run(["--no-extras", "--imu", "--nseq", "1", "--steps", "200", "--warmup", "12", "--cpu-frames", "0"], 300)
            best = js["ms_per_step"] if best is None else min(best, js["ms_per_step"])
        if isinstance(extras.get("imu"), dict):
            extras["imu"]["single_sequence_ms_per_frame"] = best
    except Exception as e:
        if isinstance(extras.get("imu"), dict):
            extras["imu"]["single_sequence_error"] = f"{type(e).__name__}: {e}"[:200]
    return extras


def launch_ranks(n, argv, rank0_stdout):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process becomes the launcher — N copies of this
    command, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, what
    torch.distributed.run would set), rank 0's stdout is this process's stdout (the ONE line), the other ranks' goes to stderr.
    Fails loudly (exit code != 0, nothing on stdout) when the box has fewer than N devices: a `--gpus 8` command never turns
    into a one-GPU number.  BENCH_BACKEND=gloo lets the ranks share the devices there are (a dry run of the control flow)."""
    import socket
    import subprocess
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    stub = os.environ.get("BENCH_STUB_DEVICE") == "1"
    if not stub:
        import torch
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < 1:
            print(f"bench.py: --gpus {n} but this box has no usable GPU; there is no CPU path", file=sys.stderr)
            return 2
        if backend == "nccl" and ndev < n:
            print(f"bench.py: --gpus {n} but only {ndev} device(s) are visible: one rank per GPU over RCCL needs {n}.  Nothing was "
                  f"measured.  (BENCH_BACKEND=gloo runs {n} ranks on the devices there are, as a dry run of the control flow.)", file=sys.stderr)
            return 2
    elif backend == "nccl":
        print("bench.py: BENCH_STUB_DEVICE=1 is a CPU dry run of the launcher and needs BENCH_BACKEND=gloo", file=sys.stderr)
        return 2
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, _usable_cores() // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=rank0_stdout if r == 0 else sys.stderr, stderr=sys.stderr))
    # a rank that dies takes the others with it (they would wait for it in the next collective until the store times out)
    rc, alive = 0, list(procs)
    while alive:
        for p_ in list(alive):
            code = p_.poll()
            if code is None:
                continue
            alive.remove(p_)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {procs.index(p_)} ended with exit code {code}; stopping the other ranks", file=sys.stderr)
                for q_ in alive:
                    q_.terminate()
        time.sleep(0.05)
    return rc


def main():
    # ONE JSON line on stdout, whatever the libraries underneath print: RCCL writes a version banner to the C-level stdout of
    # the process when its first communicator comes up.  File descriptor 1 is pointed at stderr for the rest of the run and
    # Python's own sys.stdout (the line at the end) keeps the real one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(real_stdout, "w")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--config", default="full", choices=["full", "stage_a", "tum_undistort"],
                    help="full = BASELINE configs[2]/[4] (the default line); stage_a = configs[1]; tum_undistort = configs[3]")
    ap.add_argument("--nseq", type=int, default=1024, help="independent sequences per GPU (all contexts together)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="edgehip contexts (= HIP streams) the sequences are split over: kernels of different contexts "
                         "run concurrently, which hides the serial LM-step kernels and launch tails of one context "
                         "behind the bandwidth-bound kernels of the other")
    ap.add_argument("--pool", type=int, default=24, help="rendered frames in the HBM pool")
    ap.add_argument("--cpu-frames", type=int, default=100, help="frames of the CPU-baseline sample (0 = skip the CPU legs)")
    ap.add_argument("--input", default="distinct", choices=["distinct", "pool"],
                    help="where the resident frames of the timed region live: distinct = every sequence reads its OWN copy of its frames "
                         "(sequences x pool frames x 3 B per pixel: 26.6 GB at the defaults, far beyond L2 + Infinity Cache, so stage A's "
                         "input really comes from HBM); pool = all sequences read one shared pool of --pool frames (26 MB: cache-served)")
    ap.add_argument("--extras", action="store_true",
                    help="also measure (into bench_extras.json, never into the one line): the heterogeneous batch with its teacher-forced "
                         "replay, the PCIe-inclusive legs, and the other BASELINE configurations (stage_a, tum_undistort, ImuMode=2) each "
                         "in a process of its own.  Minutes of wall time; off by default")
    ap.add_argument("--overlap", action="store_true",
                    help="EDGEHIP_OVERLAP=1: stage A of frame k+1 under stages B/C of frame k (two streams per context). "
                         "Faster, but per-kernel HIP-event times (the roofline object) stop being attributable, so off by default")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the node-saturating CPU leg (SURVEY.md section 8d mode iii; -1 = usable cores // 3 "
                         "sequences, each with the reference's two compute threads; 0 = skip)")
    ap.add_argument("--imu", action="store_true",
                    help="ImuMode=2 (what the shipped GlobalConfig_EuRoC runs): the IMU branch of the tracker — gyro pre-rotation, "
                         "Minimizer_V, ExtRotVel, BiasCorrect, scale filter, gravity-aligned pose — batched on the device; the "
                         "integrated IMU data of every frame interval is synthesised from the known camera motion")
    ap.add_argument("--tracker-f32", action="store_true",
                    help="a SECOND configuration, never the headline: the tracker's float instantiation, Minimizer_RV<float> / TryVelRot<float> "
                         "(global_tracker.cpp:824; what the reference runs when built with USE_NE10, rebvo_second_t.cpp:339-343) — "
                         "edgehip_set_tracker_precision(ctx, 32).  Pose RMSE and the CPU baseline are then taken against the reference's own "
                         "float instantiation; the line's dtype says so")
    ap.add_argument("--no-extras", action="store_true", help="skip the small-batch sweep and the plugin-surface legs as well (profiling runs)")
    ap.add_argument("--dataset", default=os.environ.get("REBVO_DATASET_DIR") or os.environ.get("REBVO_EUROC_DIR") or os.environ.get("REBVO_TUM_DIR"),
                    help="a mounted EuRoC sequence (the directory that holds mav0/cam0/data.csv, 752x480: the default line) or TUM "
                         "sequence (rgb.txt, 640x480: --config tum_undistort): its images, read by the library's own DataSetCam, are "
                         "the frame pool instead of the synthetic billboards — sequence s of the batch starts at its own frame of the "
                         "list; the pose check runs the CPU reference on the same files.  Also REBVO_DATASET_DIR / REBVO_EUROC_DIR / "
                         "REBVO_TUM_DIR.  Absent or not a data set: the synthetic scenes, silently")
    ap.add_argument("--dataset-frames", type=int, default=96, help="images of the list that make the HBM-resident pool")
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the two rocprofv3 counter passes of this command (N = 1 only) that make roofline.traffic an observation of "
                         "this run; the committed counters of profiles/pmc_latest.json are used instead, under their source stamp")
    args = ap.parse_args()
    user_no_extras = bool(args.no_extras)   # (--imu / --tracker-f32 switch the extras off further down: the counter passes go by what was asked for)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around this command (the driver runs `python bench.py --gpus N` as it runs `--gpus 1`): be the launcher
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], sys.stdout))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # never a number for another rank count than the command names
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment: launch `--gpus N` with N ranks "
              "(torch.distributed.run --nproc-per-node N), or without a launcher (bench.py then starts its own N ranks)", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    from rebvo_amd import edgehip, shard, synth

    # BENCH_STUB_DEVICE=1 (tests/test_bench_launcher_cpu.py): the control flow of an N-rank run on a box without a GPU, with
    # tests/stub_device.py in the place of the device context.  Nothing is measured; the line says so.
    stub = os.environ.get("BENCH_STUB_DEVICE") == "1"
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev = "cpu" if stub else "cuda"
    if stub:
        from tests import stub_device
        import types
        if backend == "nccl" or args.config != "full" or args.imu:
            raise SystemExit("BENCH_STUB_DEVICE=1: the dry run covers the default configuration over BENCH_BACKEND=gloo only")
        edgehip_dev = types.SimpleNamespace(EdgeHip=lambda params, nseq, nslots, device: stub_device.EdgeHip(params, nseq, nslots, rank))
        torch.cuda.synchronize = lambda *a, **k: None
        args.input, args.cpu_frames, args.no_extras, args.no_roofline_events = "pool", 0, True, True
    else:
        edgehip_dev = edgehip
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
        # BENCH_BACKEND=gloo is a dry run of the multi-rank control flow on a box with fewer GPUs than ranks (ranks share
        # devices); the real thing is nccl (= RCCL), one rank per GPU
        if backend != "nccl":
            local_rank %= torch.cuda.device_count()
        elif local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} wants device {local_rank} but only {torch.cuda.device_count()} are visible")
        torch.cuda.set_device(local_rank)
    # One rank, full path: the N > 1 code path is taken all the same (BENCH_FORCE_MOVER=0 turns it off) — a one-rank process
    # group over RCCL, shard.NavMover on its own communicator and thread, barriers, the max-over-ranks all-reduce — so that the
    # line a 1-GPU run prints has exercised what an 8-GPU run does, minus the peers (`nav_gather` in the line).  A box on which
    # RCCL does not come up still measures: `nav_gather` then says why.
    force_mover = world == 1 and args.config == "full" and os.environ.get("BENCH_FORCE_MOVER", "1") != "0" and not stub
    dist_on = world > 1
    dist_note = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif force_mover:
        try:
            import socket
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend, rank=0, world_size=1)
            dist_on = True
        except Exception as ex:
            dist_note = f"unavailable: {type(ex).__name__}: {str(ex)[:120]}"

    if args.overlap:
        os.environ["EDGEHIP_OVERLAP"] = "1"
    tum = args.config == "tum_undistort"
    w, h = (640, 480) if tum else (W, H)
    n_px = w * h
    params = edgehip.tum_params(w, h, use_undistort=1) if tum else edgehip.euroc_params(w, h)
    radius = params.search_range
    C = max(1, args.contexts)
    B, K, Wm = args.nseq // C, args.steps, args.warmup   # B = sequences per context
    # every rank checks a sample of its own sequences against the CPU reference (pose RMSE); rank 0 also times the reference
    cpu_any = args.cpu_frames > 0
    cpu_legs = cpu_any and not args.imu
    cpu_legs_imu = cpu_any and args.imu and rank == 0 and world == 1
    if args.imu:
        args.no_extras = True   # the CPU legs and the other batch shapes are those of the ImuMode=0 line
        global ONE_PASS_MATCHING
        ONE_PASS_MATCHING = False   # ExtRotVel reads the forward matches between FordwardMatch and directed_matching: three kernels

    def to_pool(frames_list):
        """HBM-resident frame pool; stage A reads each sequence's frame in place (edgehip_bind_rgb_indexed: no gather
        copy, like ConvertRGB2BW reading the camera buffer).  16 B of slack: pixels are fetched as aligned 8-byte words."""
        host = np.stack(frames_list)
        t = torch.empty(host.size + 16, dtype=torch.uint8, device=dev)
        t[:host.size] = torch.from_numpy(host.reshape(-1)).to(dev)
        return t

    # ---- frame pool, resident in HBM: a mounted data set when there is one, the synthetic billboards otherwise ----
    intr = dict(fx=float(params.zfx), fy=float(params.zfy), cx=float(params.ppx), cy=float(params.ppy))
    global FRAME_DT
    data_kind, data_note = "synthetic", None
    found = find_dataset(args.dataset) if args.config != "stage_a" or args.dataset else None
    if found and not args.imu:
        ds_frames, ds_t = load_dataset(found, w, h, max(8, args.dataset_frames))
        frames = list(ds_frames)
        args.pool = len(frames)
        FRAME_DT = float(np.mean(np.diff(ds_t)))
        data_kind = found[0]
        data_note = {"list": found[2], "frames_in_pool": len(frames), "mean_frame_interval_s": round(FRAME_DT, 6),
                     "first_stamp": float(ds_t[0]), "reader": "rebvo::DataSetCam of librebvohost.so (rebvo/dataset_c.h)"}
    else:
        frames = [f for f, _, _ in synth.billboard_sequence(w, h, args.pool, seed=11 + rank, **intr)]
    pool = to_pool(frames)
    torch.cuda.synchronize()
    if stub:
        data_kind = "stub"
    # every sequence starts at its own phase of the pool
    offs = np.arange(B * C, dtype=np.int64) % (2 * (args.pool - 1))
    # --input distinct: sequence s reads frame i of the pool from ITS OWN copy (frame s * pool + i of a B*C*pool-frame array), as B*C
    # cameras' buffers would lie in memory: the same pixels as the shared pool (so the CPU reference replays the same frames), at
    # addresses nobody else touches — at the defaults 26.6 GB are swept per pass over the pool, against 256 MB of Infinity Cache
    fb = int(frames[0].size)
    in_pool, in_frames, in_note = pool, args.pool, f"one shared pool of {args.pool} RGB24 frames ({args.pool * fb / 1e6:.0f} MB: cache-served)"
    seq_base = np.zeros(B * C, dtype=np.int64)
    if args.input == "distinct" and B * C * args.pool * fb <= 96e9 and B * C * args.pool < 2 ** 31:
        nS = B * C
        big = torch.empty(nS * args.pool * fb + 16, dtype=torch.uint8, device="cuda")
        big[:nS * args.pool * fb].view(nS, args.pool * fb).copy_(pool[:args.pool * fb].view(1, -1).expand(nS, -1))
        torch.cuda.synchronize()
        in_pool, in_frames = big, nS * args.pool
        del big
        seq_base = np.arange(nS, dtype=np.int64) * args.pool
        in_note = (f"every sequence reads its own RGB24 copy of its frames: {nS} x {args.pool} frames = {nS * args.pool * fb / 1e9:.1f} GB "
                   "resident, read in place by stage A")

    def index_of(k):
        return seq_base + tri_v(k + offs, args.pool)
    imu_params = imu_of = None
    if args.imu:
        # Integrated IMU data (rebvo::IntegratedImuData) of every transition between two pool frames, from the known motion:
        # the camera turns by exp(tw_rot) (points) from pool frame i to i+1, i.e. the body rate is -tw_rot / dt; a constant
        # gyro bias on top; the accelerometer sees gravity in the camera frame of the frame it arrives at.
        tw = synth.smooth_trajectory(args.pool, 13)
        poses = [R for _, R, _ in synth.billboard_sequence(8, 8, args.pool, seed=11 + rank)]   # the rotations only (tiny render)
        bias, dt_f, nsamp = np.array([0.004, -0.002, 0.003]), 0.05, 10
        imu_params = edgehip.euroc_imu_params(init_bias_frame_num=3)

        def imu_record(i_from, i_to):
            rot = tw[i_from, 3:] if i_to > i_from else -tw[i_to, 3:]
            giro = -rot / dt_f + bias
            r = edgehip.ImuIntegrated()
            r.n, r.dt = nsamp, nsamp * dt_f / nsamp
            r.Rot[:] = synth._so3_exp(giro * dt_f).reshape(-1)
            r.giro[:] = giro
            acc = -(poses[i_to] @ np.array([0.0, 9.8, 0.0]))
            r.acel[:] = acc
            r.cacel[:] = acc
            return r
        trans = {}
        for i in range(args.pool):
            for j in (i - 1, i, i + 1):
                if 0 <= j < args.pool:
                    trans[(i, j)] = imu_record(i, j) if i != j else imu_record(i, min(i + 1, args.pool - 1))

        def imu_of(k):
            return [trans[(tri(k - 1 + o, args.pool) if k > 0 else tri(k + o, args.pool), tri(k + o, args.pool))] for o in offs]
    rp = Replay(edgehip_dev, params, B * C, in_pool, in_frames, index_of, local_rank, C, imu_params=imu_params, imu_of=imu_of)
    ehs, eh = rp.ehs, rp.ehs[0]   # eh: the context whose streams carry the HIP-event profiler
    f32 = bool(args.tracker_f32)
    if f32:
        if args.imu or args.config == "stage_a" or stub:
            raise SystemExit("--tracker-f32 is the ImuMode = 0 tracker's float instantiation (full path only)")
        for e in ehs:
            e.set_tracker_precision(32)
        args.no_extras = True

    # ============================ --config stage_a: DoG + KeyLine extraction alone (configs[1]) ============================
    if args.config == "stage_a":
        def stage_a_step(k):
            idx = np.ascontiguousarray(index_of(k), dtype=np.int32)
            for ci, e in enumerate(ehs):
                s = k % 3
                e.bind_rgb_indexed(s, in_pool.data_ptr(), in_frames, idx[ci * B:(ci + 1) * B])
                e.stage_a(s)
        for k in range(Wm):
            stage_a_step(k)
        rp.sync()
        eh.profile_enable(True)
        eh.profile_select(None)
        # ---- timed region (stage A only): EXACTLY K steps between barriers ----
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for k in range(Wm, Wm + K):
            stage_a_step(k)
        rp.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        # ---- end of timed region ----
        prof = {g: (ms, calls) for g, (ms, calls) in eh.profile_read().items() if calls}
        kn_mean = float(np.mean([v for e in ehs for v in e.get_kn((Wm + K - 1) % 3)]))
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        if rank == 0:
            fbytes = 3 * n_px + 4 * n_px + 168 * kn_mean       # stage A
            fps = B * C * world * K / dt
            dom = max(prof, key=lambda g: prof[g][0])
            per_launch = prof[dom][0] * 1e-3 / prof[dom][1]
            ab = algorithmic_bytes(dom, kn_mean, n_px, radius, B)
            cpu = None
            if cpu_legs:
                from oracle import oracle
                orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
                tr, lr, ts = orc.p.detector_thresh, 0, []
                for k in range(8 + min(args.cpu_frames, 100)):
                    t1 = time.perf_counter()
                    _, tr, lr = orc.stage_a(k % 8, frames[tri(k, args.pool)], tr, lr)
                    ts.append(time.perf_counter() - t1)
                ts = np.array(ts[8:])
                cpu = {"value": round(fbytes / float(ts.mean()) / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "reference",
                       "sample": f"{len(ts)} frames of stage A (sspace::build + edge_finder::detect) on 1 core of {_usable_cores()} "
                                 f"({_cpu_model()})", "ms_per_frame": round(float(ts.mean()) * 1e3, 3),
                       "median_ms": round(float(np.median(ts)) * 1e3, 3), "p95_ms": round(float(np.percentile(ts, 95)) * 1e3, 3)}
            print(json.dumps({
                "metric": "HBM GB/s (DoG + edge_finder KeyLine extraction) 752x480", "value": round(fbytes * fps / world / 1e9, 2),
                "unit": "GB/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(dt / K * 1e3, 4),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 scale space / f64 plane fit",
                "data": "synthetic",
                "config": {"workload": "BASELINE configs[1]: stage A only (RGB->grey, box-filter scale space, DoG, KeyLine extraction, "
                                       "join_edges, threshold control), 752x480, GlobalConfig_EuRoC detector parameters; value = "
                                       "SURVEY 8(d) stage-A bytes (3N + 4N + 168 kn) x frames/s per GPU",
                           "sequences_per_gpu": B * C, "frames_per_step": B * C * world, "keylines_per_frame": round(kn_mean, 1),
                           "frames_per_s": round(fps, 1), "algorithmic_MB_per_frame": round(fbytes / 1e6, 3)},
                "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ab / per_launch / 1e9, 2), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(ab / per_launch / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                             "launch_us": round(per_launch * 1e6, 2), "algorithmic_bytes_per_launch": int(ab),
                             "launches_timed": prof[dom][1]},
                "stage_a_hbm_frac": round(fbytes * fps / world / 1e9 / HBM_PEAK_GBS, 5),
                "cpu_baseline": cpu,
                "kernel_us_per_step": {g: round(ms / K * 1e3, 1) for g, (ms, calls) in prof.items()}}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ============================ the full path ============================
    for e in ehs:
        e.set_nav_log(Wm + K)   # the whole replay from frame 0: the parity leg looks for the first frame a sequence leaves the reference on

    def barrier():
        rp.sync()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()

    # ---- warmup (also finds the dominant kernel group with the built-in HIP-event profiler) ----
    prof_steps = min(4, max(1, Wm // 4))
    for k in range(Wm - prof_steps):
        rp.step(k)
    rp.sync()
    eh.profile_enable(True)
    eh.profile_select(None)
    for k in range(Wm - prof_steps, Wm):
        rp.step(k)
    prof = eh.profile_read()
    eh.profile_enable(False)
    groups = {g: (ms, calls) for g, (ms, calls) in prof.items() if calls}
    dominant = max(groups, key=lambda g: groups[g][0]) if groups else None
    breakdown = {g: round(ms / prof_steps * 1e3, 1) for g, (ms, calls) in groups.items()}  # us per step

    # ---- timed region: EXACTLY K steps between barriers ----
    if dominant and not args.no_roofline_events:
        eh.profile_select([dominant])
        eh.profile_enable(True)
    # N > 1: the nav records of a step travel to rank 0 in blocks of `blk` steps on a side stream (RCCL gather issued
    # while the next block is being computed; two buffers), the last block after the timed region's last step
    mover = shard.NavMover(world, rank, backend, device=local_rank if backend == "nccl" else None) if dist_on else None
    nav_gather = dist_note
    blk = max(1, K // 4)
    barrier()
    t0 = time.perf_counter()
    posted = 0
    for k in range(Wm, Wm + K):
        rp.step(k)
        done = k - Wm + 1
        if mover and (done % blk == 0 or done == K):   # the block just enqueued: the mover's thread waits for it, this one does not
            try:
                for ci, e in enumerate(ehs):
                    mover.post(e, Wm + posted, done - posted, [(rank * C + ci) * B + s for s in range(B)])
            except Exception as ex:     # the records are a by-product: a broken transport must not take the measurement down
                nav_gather, mover = f"failed: {str(ex)[:120]}", None
            posted = done
    if mover:
        try:
            mover.finish(timeout=120)   # every record has reached rank 0 inside the timed region; all but the last block under compute
            nav_gather = "ok"
        except Exception as ex:
            nav_gather = f"failed: {str(ex)[:120]}"
    barrier()
    dt = time.perf_counter() - t0
    # ---- end of timed region ----
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    nav_gather_info = None
    if mover is not None and nav_gather == "ok" and rank == 0:
        # what arrived on rank 0 is what the contexts logged (every block, every rank's sequences, in order)
        got = np.concatenate([b[0] for b in mover.blocks], axis=0) if world == 1 and C == 1 else None
        nav_gather_info = {"backend": "rccl" if backend == "nccl" else backend, "ranks": world, "blocks": len(mover.blocks),
                           "records": int(sum(b.shape[0] * b.shape[1] * b.shape[2] for b in mover.blocks)),
                           # RCCL: every block went device log -> device tensor -> communicator (edgehip_read_nav_log_device), no host bounce
                           "device_to_wire": bool(backend == "nccl" and mover.device_path_blocks == len(mover.blocks) > 0)}
        if world > 1:
            # every rank's every sequence, every timed step, once: (rank, sequence id, frame) of what arrived
            allrec = np.concatenate([b for b in mover.blocks], axis=1)          # [world, K, B, 16] (C == 1) or per-context blocks
            want_ranks = sorted(set(int(x) for x in allrec[..., 14].reshape(-1)))
            nav_gather_info["ranks_seen"] = want_ranks
            nav_gather_info["every_rank_delivered"] = want_ranks == list(range(world))
        if got is not None:
            direct = shard.nav_records(eh.read_nav_log_array(Wm, K), rank, list(range(B)))
            nav_gather_info["equals_device_log"] = bool(got.shape == direct.shape and np.array_equal(got, direct))
    if dist_on and world == 1:
        dist.destroy_process_group()
        dist_on = False

    dom_ms, dom_calls = (0.0, 0)
    if dominant and not args.no_roofline_events:
        dom_ms, dom_calls = eh.profile_read()[dominant]
        eh.profile_enable(False)
    # sequences of context 0 compared with the CPU reference below: N = 1 every 32nd (32 of 1024) and the last one; N > 1 four
    # (and the last) on EVERY rank, so that a --gpus 8 line carries a pose check of all eight shards
    n_check = 32 if world == 1 else 4
    check_seqs = sorted(set(range(0, B, max(1, B // n_check))) | {B - 1})
    gpu_traj = None
    log_all = None
    if cpu_legs:
        log_all = eh.read_nav_log_array(0, Wm + K)
        log0 = log_all[Wm:]
        gpu_traj = {s: _traj_of_log(log0, s) for s in check_seqs}
    elif cpu_legs_imu:   # ImuMode > 0: the gravity-aligned pose, metric velocity and rotation of NavData (rebvo_second_t.cpp:519-606)
        log0 = eh.read_nav_log_array(Wm, K)
        gpu_traj = {s: [(log0[k, s]["Pos"].copy(), log0[k, s]["Pose"].reshape(3, 3).copy(), log0[k, s]["Vel"].copy(),
                         log0[k, s]["RotLie"].copy()) for k in range(K)] for s in check_seqs}
    last = [n for e in ehs for n in e.read_nav()]
    kn_mean = float(np.mean([n.kn for n in last]))
    kn_timed = float(np.mean((log_all[Wm:] if log_all is not None else eh.read_nav_log_array(Wm, K))["kn"])) if rank == 0 else kn_mean   # over every timed frame, context 0
    ok = int(sum(n.estimation_ok for n in last))
    evals = last[0].minimizer_evals

    # ---- pose RMSE vs the CPU reference (BASELINE.json's metric): every rank checks its own sample, rank 0 holds the sum ----
    pose = None
    parity = None
    oparams = None
    kind = None
    if cpu_legs:
        try:
            from oracle import oracle
            oparams = oracle.tum_params(w, h, use_undistort=1) if tum else oracle.euroc_params(w, h)
            kind = "reference" if oracle.available("ref") else ("port" if oracle.available("port") else None)
            if kind == "reference":
                parity, cpu_trajs = wide_parity(log_all, check_seqs, lambda s_: [tri(k + int(offs[s_]), args.pool) for k in range(Wm + K)],
                                                np.stack(frames), ("tum" if tum else "euroc", w, h, FRAME_DT, f32), Wm,
                                                max(1, (_usable_cores() - 1) // world),
                                                **(dict(tol_rel=1e-4, tol_abs=1e-7) if f32 else {}))
                if f32 and parity and rank == 0:
                    # the parity statement of the float configuration: teacher-forced (oracle/teacher.py) — the reference's float state
                    # injected before every frame — on three sequences of the batch, every frame within a few float ulps and the
                    # discrete counts identical; and, as the yardstick for a free-running float trajectory, how far the reference's own
                    # float and double instantiations are apart on the same frames at the last frame
                    from oracle import teacher
                    tf_out, tf_max, tf_frames, yard, tf_list = 0, 0.0, 0, [], []
                    tf_seqs = check_seqs[:3]
                    for s_ in tf_seqs:
                        o32 = oracle.Oracle("ref", oparams)
                        o32.set_tracker_f32(1)
                        e1 = edgehip.EdgeHip(params, nseq=1, nslots=3, device=local_rank)
                        e1.set_tracker_precision(32)
                        tfr = teacher.teacher_forced_replay(e1, o32, lambda k, s_=s_: frames[tri(k + int(offs[s_]), args.pool)], Wm + K, dt=FRAME_DT,
                                                            tol_rel=2e-5, tol_abs=2e-8)
                        e1.close()
                        o32.close()
                        tf_out += len(tfr["outside_tolerance"])
                        tf_list += [{"sequence": int(s_), "frame": o_["frame"], "dV": o_.get("dV"), "dW": o_.get("dW"), "kn_ok_klm": o_.get("kn_ok_klm")}
                                    for o_ in tfr["outside_tolerance"]]
                        tf_frames += tfr["frames"]
                        for dv_, dw_, r_ in zip(tfr["dV"][1:], tfr["dW"][1:], tfr["ref"][1:]):
                            tf_max = max(tf_max, max(dv_, dw_) / (np.linalg.norm(r_[2]) + np.linalg.norm(r_[3]) + 1e-30))
                        o64 = oracle.Oracle("ref", oparams)
                        p64 = None
                        for k in range(Wm + K):
                            _, n64 = o64.process_frame(frames[tri(k + int(offs[s_]), args.pool)], FRAME_DT * k)
                            p64 = np.array(n64.Pos[:])
                        o64.close()
                        yard.append(float(np.linalg.norm(p64 - tfr["ref"][-1][0])))
                    parity["teacher_forced"] = {"sequences": [int(x) for x in tf_seqs], "frames": tf_frames, "frames_outside_tolerance": tf_out,
                                                "max_rel_dX": tf_max, "tolerance": "every frame |dV|,|dW| <= 2e-5*(|V|+|W|)+2e-8, identical kn / klm / EstimationOK",
                                                "reference_float_vs_reference_double_position_at_last_frame": yard, "outside": tf_list[:12]}
                if f32 and parity:
                    parity["note"] = ("float tracker against the reference's float instantiation: per frame the two agree to a few float ulps "
                                      "(teacher-forced: <= 4e-6 |X|, tests/test_tracker_f32_gpu.py); a free-running float trajectory leaves the other "
                                      "at the first Levenberg-Marquardt / initialisation decision that hangs on the last bits of a float sum — the "
                                      "reference's own float and double instantiations part the same way — so `departures_elsewhere` counts such "
                                      "decisions here, not defects")
            elif kind:
                cpu_trajs = {s_: _cpu_traj(oracle, oparams, lambda k, s_=s_: frames[tri(k + int(offs[s_]), args.pool)], Wm, K)
                             for s_ in check_seqs[:3]}
            if kind:
                pose = pose_rmse(gpu_traj, cpu_trajs, kind)
                if pose and parity:
                    pose["free_running_parity"] = parity
                    if parity.get("teacher_forced"):
                        pose["teacher_forced"] = parity["teacher_forced"]
        except Exception as e:  # the oracle is optional test infrastructure; never fatal for the bench
            pose = {"error": f"{type(e).__name__}: {e}"[:200]}
    if world > 1 and cpu_any:
        # sum over ranks: squared errors and counts (a rank whose check failed contributes nothing and is counted in `ranks_failed`)
        good = bool(pose and "position" in pose)
        par = (pose or {}).get("free_running_parity") or {}
        vec = [pose["position"] ** 2 * pose["frames"], pose["rotation_rad"] ** 2 * pose["frames"], pose["frames"], len(pose["sequences"]),
               par.get("sequences_outside_tolerance_at_last_frame", 0), par.get("departures_elsewhere", 0),
               par.get("departures_on_knife_edge_frames", 0), 0.0] if good else [0.0] * 7 + [1.0]
        t_ = torch.tensor(vec, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t_, op=dist.ReduceOp.SUM)
        v_ = [float(x) for x in t_.tolist()]
        if v_[2] > 0:
            pose = {"position": float(np.sqrt(v_[0] / v_[2])), "rotation_rad": float(np.sqrt(v_[1] / v_[2])), "frames": int(v_[2]),
                    "sequences_checked": int(v_[3]), "ranks": world, "ranks_failed": int(v_[7]),
                    "vs": f"CPU reference on the same frames, {n_check + 1} sequences of every rank's shard",
                    "free_running_parity": {"sequences_checked": int(v_[3]), "sequences_outside_tolerance_at_last_frame": int(v_[4]),
                                            "departures_elsewhere": int(v_[5]), "departures_on_knife_edge_frames": int(v_[6])}}

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = B * C * world * K / dt
    # ---- per-kernel attribution: four more steps (frames right behind the timed ones, i.e. of the same cost) with the HIP-event
    # profiler on every group — not part of the timed region, and not the early warm-up frames whatever --warmup is ----
    prof_steps = 4
    eh.profile_select(None)
    eh.profile_enable(True)
    for k in range(Wm + K, Wm + K + prof_steps):
        rp.step(k)
    prof = eh.profile_read()
    eh.profile_enable(False)
    groups = {g: (ms, calls) for g, (ms, calls) in prof.items() if calls}
    breakdown = {g: round(ms / prof_steps * 1e3, 1) for g, (ms, calls) in groups.items()}  # us per step
    # ---- roofline of the dominant kernel group, and of all of them ----
    # `frac` = compulsory bytes of what is launched / launch time / 8 TB/s; `frac_survey_formula` = the same with SURVEY 8(d)'s
    # per-unit figure where that counts bytes a launch does not have to move; `frac_on_traffic` = the rate at which the kernel
    # really moves bytes (committed PMC passes — used only when they were taken with THIS library's sources, else null);
    # `issue_frac` = share of the kernel's cycles in which its SIMDs issue VALU work (committed SQ counters, same rule): a kernel
    # priced only against a roofline it is not bound by tells nobody anything.
    under_profiler = any("rocprof" in os.environ.get(v, "") for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    if (world == 1 and not stub and not args.no_live_pmc and not user_no_extras and not os.environ.get("BENCH_LIVE_PMC_CHILD")
            and not under_profiler and args.config == "full"):
        child = [a for a in sys.argv[1:] if a != "--extras"] + ["--gpus", "1", "--nseq", str(B * C), "--steps", str(K), "--warmup", str(Wm),
                                                                "--cpu-frames", "0", "--no-extras", "--no-roofline-events", "--no-live-pmc"]
        torch.cuda.synchronize()
        live_pmc_passes(child, B)
    elif not LIVE_PMC["used"]:
        LIVE_PMC["note"] = "switched off (--no-live-pmc / --no-extras / N > 1 / a profiler around this process)"
    calib, calib_src = fetch_calibration()
    kn_pmc = pmc_kn() or kn_mean
    issue = issue_fracs()

    def roof_of(g, ms, calls, kn):
        ab = algorithmic_bytes(g, kn, n_px, radius, B)
        if not ab or not calls:
            return None
        per = ms * 1e-3 / calls
        r_ = {"launch_us": round(per * 1e6, 1), "achieved_GBs": round(ab / per / 1e9, 1), "frac": round(ab / per / 1e9 / HBM_PEAK_GBS, 4),
              "algorithmic_bytes_per_launch": int(ab)}
        sb = survey_bytes(g, kn, n_px, radius, B)
        if sb != ab:
            r_["frac_survey_formula"] = round(sb / per / 1e9 / HBM_PEAK_GBS, 4)
        tr, _ = calibrated_traffic(g, B, kn_pmc, calib)
        ab_pmc = algorithmic_bytes(g, kn_pmc, n_px, radius, B)
        if tr and ab_pmc:   # counters taken at another KeyLine count: the ratio to the algorithmic bytes there, applied here
            r_["traffic"] = int(tr / ab_pmc * ab)
            r_["frac_on_traffic"] = round(tr / ab_pmc * ab / per / 1e9 / HBM_PEAK_GBS, 4)
        if g in issue and not (f32 and not LIVE_PMC.get("issue") and g.startswith("B.try_velrot")):   # (the committed SQ passes are the fp64 kernels')
            r_["issue_frac"] = issue[g]
        return r_
    roof = None
    if dominant and dom_calls:
        # bytes of THESE launches / time of THESE launches: the KeyLine count of the timed frames (kn_timed)
        rd = roof_of(dominant, dom_ms, dom_calls, kn_timed)
        if rd:
            roof = {"bound": "hbm", "kernel": dominant, "achieved": rd["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": rd["frac"], "frac_on_traffic": rd.get("frac_on_traffic"), "traffic": rd.get("traffic"),
                    "launch_us": rd["launch_us"], "algorithmic_bytes_per_launch": rd["algorithmic_bytes_per_launch"],
                    "launches_timed": dom_calls, "issue_frac": rd.get("issue_frac")}
            if "frac_survey_formula" in rd:
                roof["frac_survey_formula"] = rd["frac_survey_formula"]
            roof["traffic_live"] = bool(LIVE_PMC["used"]) and roof["traffic"] is not None
            roof["issue_live"] = bool(LIVE_PMC.get("issue")) and roof.get("issue_frac") is not None
            if roof["traffic"] is None:
                roof["traffic_note"] = pmc_stamp_note()
    roof_all = {}
    for g, (ms, calls) in groups.items():
        rg = roof_of(g, ms, calls, kn_mean)
        if rg:
            rg["launches_per_step"] = calls // prof_steps
            roof_all[g] = rg
    # whole-frame algorithmic bytes, SURVEY.md §8(d) formulas with the measured kn and evaluation count
    r = radius
    frame_bytes = (3 * n_px + 4 * n_px + 168 * kn_mean) + (8 * n_px + 8 * 2 * r * kn_mean + evals * 84 * kn_mean) + \
                  ((4 * 40 + 2 * 168) * kn_mean + 100 * kn_mean + 64 * kn_mean + 160 * kn_mean)
    if tum:
        frame_bytes += 36 * n_px    # the undistortion map read
    rp.close()
    rp.pool_t = None
    del in_pool
    torch.cuda.empty_cache()

    # ---- CPU baseline: the reference's own code on the host cores of this box (rank 0): one core, the reference's own two-thread
    # overlap, node-saturating.  N > 1: the one-core leg only, on a shorter sample (the other ranks wait at the last barrier) ----
    cpu = None
    if cpu_legs and kind:
        try:
            from oracle import oracle
            ncores, model = _usable_cores(), _cpu_model()
            nfr_cpu = args.cpu_frames if world == 1 else min(args.cpu_frames, 60)
            if kind == "reference":
                host_pool = np.stack(frames)
                idx = [tri(k, args.pool) for k in range(10 + nfr_cpu)]   # the first 10: first touch of the ring + MKL init
                modes = {}
                for name, th in (("serial_1_core", 1), ("reference_threads_2_cores", 2))[:2 if world == 1 else 1]:
                    orc = oracle.Oracle("ref", oparams)
                    if f32:
                        orc.set_tracker_f32(1)
                    done, _ = orc.run_sequence(host_pool, idx, dt=FRAME_DT, threads=th)
                    orc.close()
                    modes[name] = frame_stats(done, 10)
                    modes[name]["cores"] = th
                cpu = dict(modes["serial_1_core"])
                cpu.update({"cores": 1, "kind": kind, "cpu_model": model, "usable_cores": ncores,
                            "sample": f"{nfr_cpu} frames of sequence 0 (same {w}x{h} frames), reference mtracklib stage A + B/C back to "
                                      f"back on 1 of {ncores} host cores ({model})",
                            "modes": modes})
            else:
                orc = oracle.Oracle("port", oparams)
                ts = []
                for k in range(10 + nfr_cpu):
                    _, nav = orc.process_frame(frames[tri(k, args.pool)], FRAME_DT * k)
                    ts.append(nav.dtp0 + nav.dtp1)
                done = np.cumsum(ts)
                cpu = frame_stats(done, 10)
                cpu.update({"cores": 1, "kind": kind, "cpu_model": model, "usable_cores": ncores,
                            "sample": f"{nfr_cpu} frames of sequence 0, restatement oracle on 1 core"})
        except Exception as e:  # the oracle is optional test infrastructure; never fatal for the bench
            cpu = {"value": None, "error": str(e)[:200]}
        if cpu and f32:
            cpu["sample"] = cpu["sample"].replace("reference mtracklib", "reference mtracklib with Minimizer_RV<float> (its USE_NE10 tracker)")
        if cpu and cpu.get("value") and cpu.get("kind") == "reference" and args.cpu_procs and not tum and world == 1 and not f32:
            # node-saturating mode (SURVEY 8d iii): P independent sequences in parallel processes, each with the
            # reference's two compute threads (its third thread only ships results)
            try:
                import multiprocessing as mp
                ncpu = _usable_cores()
                P = args.cpu_procs if args.cpu_procs > 0 else max(1, ncpu // 3)
                nfr = max(30, args.cpu_frames // 3)
                with mp.get_context("spawn").Pool(P) as pool_:   # spawn: never fork a process that holds a HIP context
                    res = pool_.map(_cpu_worker, [(cpu["kind"], nfr, args.pool, 11 + i, 2, w, h) for i in range(P)])
                cpu["modes"]["node_saturating"] = {
                    "value": round(P * nfr / max(res), 1), "unit": "frames/s", "processes": P, "threads_per_process": 2,
                    "cores": min(ncpu, 2 * P), "frames": P * nfr,
                    "sample": f"{P} sequences x {nfr} frames, one process each with the reference's two compute threads "
                              f"(slowest process {max(res):.2f} s)"}
            except Exception as e:
                cpu["modes"]["node_saturating"] = {"value": None, "error": str(e)[:200]}

    if cpu_legs_imu:
        # ImuMode > 0: the same two things for the IMU branch — pose RMSE of three sequences of the batch against the reference's
        # own ImuMode > 0 frame order on the same frames and IMU records, and that code timed on this box's host cores.  One
        # process per sequence (the reference's filter histories are process-wide statics); the harness runs the branch on one
        # thread (the reference's FirstThr / SecondThread split is timed on the ImuMode = 0 line).
        try:
            import multiprocessing as mp
            from oracle import oracle
            if not oracle.available("ref"):
                raise RuntimeError("oracle/_ref not built")
            host_pool = np.stack(frames)
            rows = lambda rec: np.array([rec.n, rec.dt] + list(rec.Rot) + list(rec.giro) + list(rec.acel) + list(rec.comp) +
                                        list(rec.dgiro) + list(rec.cacel), dtype=np.float64)
            imu_over = dict(init_bias_frame_num=3)

            def job_of(s_, nfr):
                o = int(offs[s_])
                idx = [tri(k + o, args.pool) for k in range(nfr)]
                recs = [rows(trans[(tri(k - 1 + o, args.pool) if k > 0 else idx[0], idx[k])]) for k in range(nfr)]
                return (w, h, host_pool, idx, np.array(recs), 0.05, imu_over)
            ncores, model = _usable_cores(), _cpu_model()
            nfr_t = 10 + args.cpu_frames
            jobs = [job_of(s_, Wm + K) for s_ in check_seqs] + [job_of(0, nfr_t)]
            with mp.get_context("spawn").Pool(min(len(jobs), max(1, ncores))) as pool_:
                res = pool_.map(_cpu_imu_worker, jobs)
            cpu_trajs = {}
            for s_, (secs, pos, pose_m, vel, rotlie, ok_) in zip(check_seqs, res[:len(check_seqs)]):
                cpu_trajs[s_] = {k: (pos[Wm + k], pose_m[Wm + k].reshape(3, 3), vel[Wm + k], rotlie[Wm + k]) for k in range(K)}
            pose = pose_rmse(gpu_traj, cpu_trajs, "reference (ImuMode > 0 frame order)")
            if pose:
                pose["note"] = ("V / W here = NavData::Vel (metric, gravity frame) and RotLie; Pos / Pose = the gravity-aligned, "
                                "scale-filtered pose of the IMU branch.  The scale filter's 7x7 / 11-row solves are ill-conditioned "
                                "(LAPACK SVD in the reference, Jacobi on the device): tests bound whole sequences by 1e-6..1e-5")
            secs = res[-1][0]
            done = np.cumsum(secs)
            cpu = frame_stats(done, 10)
            cpu.update({"cores": 1, "kind": "reference", "cpu_model": model, "usable_cores": ncores,
                        "sample": f"{args.cpu_frames} frames of sequence 0 (same {w}x{h} pool and IMU records) through the reference's "
                                  f"ImuMode > 0 branch on 1 of {ncores} usable host cores ({model}), next to {len(check_seqs)} more "
                                  "reference processes (the pose check); `modes` adds the node-saturating run",
                        "modes": {"serial_1_core": dict(frame_stats(done, 10), cores=1)}})
            if args.cpu_procs:
                P = args.cpu_procs if args.cpu_procs > 0 else max(1, ncores - 1)
                nfr = max(30, args.cpu_frames // 3)
                with mp.get_context("spawn").Pool(P) as pool_:
                    t_0 = time.perf_counter()
                    res2 = pool_.map(_cpu_imu_worker, [job_of(i % (B * C), 10 + nfr) for i in range(P)])
                slow = max(float(r_[0][10:].sum()) for r_ in res2)
                cpu["modes"]["node_saturating"] = {
                    "value": round(P * nfr / slow, 1), "unit": "frames/s", "processes": P, "threads_per_process": 1, "cores": min(ncores, P),
                    "frames": P * nfr, "sample": f"{P} sequences x {nfr} frames, one single-threaded reference process each (slowest {slow:.2f} s)"}
        except Exception as e:
            cpu = cpu or {"value": None}
            cpu["error"] = f"{type(e).__name__}: {e}"[:300]

    # ---- the other batch shapes (N = 1, after the timed region) ----
    sweep = hetero = None
    if world == 1 and not args.no_extras and not args.imu:
        sweep = []
        for n in (1, 8, 64):
            o = np.arange(n, dtype=np.int64) % (2 * (args.pool - 1))
            r2 = Replay(edgehip, params, n, pool, args.pool, lambda k, o=o: [tri(k + x, args.pool) for x in o], local_rank)
            k2 = 200 if n == 1 else 60
            dt2, _ = timed_replay(r2, k2, 12)
            sweep.append({"sequences_per_launch": n, "frames_per_s": round(n * k2 / dt2, 1), "ms_per_step": round(dt2 / k2 * 1e3, 4)})
            r2.close()
        sweep.append({"sequences_per_launch": B, "frames_per_s": round(value, 1), "ms_per_step": round(dt / K * 1e3, 4)})
    surface = None
    if world == 1 and not args.no_extras and not args.imu and args.config == "full" and not tum:
        try:
            surface = host_surface(params, frames, w, h)
        except Exception as e:
            surface = {"error": f"{type(e).__name__}: {e}"[:200]}
    if world == 1 and args.extras and not args.no_extras and not args.imu:
        # heterogeneous batch: six scenes with their own trajectories, every sequence at its own phase of its scene, one
        # sequence in sixteen cuts to another scene half-way through the timed region (estimation restart)
        try:
            S, PF = 6, 12
            scenes = [[f for f, _, _ in synth.billboard_sequence(w, h, PF, seed=101 + 7 * s, traj_seed=29 + s, **intr)] for s in range(S)]
            hpool = to_pool([f for sc in scenes for f in sc])
            n = B * C
            scene_of = np.arange(n) % S
            phase = (np.arange(n) // S) % (2 * (PF - 1))
            cut = (np.arange(n) % 16) == 5
            kcut = Wm + K // 2

            def hidx(k, s=None):
                sc = np.where(cut & (k >= kcut), (scene_of + 1) % S, scene_of)
                fr = np.array([tri(k + p, PF) for p in phase])
                out = sc * PF + fr
                return out if s is None else int(out[s])
            r3 = Replay(edgehip, params, n, hpool, S * PF, hidx, local_rank)
            r3.ehs[0].set_nav_log(Wm + K)
            dt3, _ = timed_replay(r3, K, Wm)
            lst = r3.ehs[0].read_nav()
            kns = np.array([x.kn for x in lst])
            hetero = {"value": round(n * K / dt3, 1), "unit": "frames/s", "ms_per_step": round(dt3 / K * 1e3, 4), "scenes": S,
                      "sequences_with_scene_cut": int(cut.sum()), "keylines_min_mean_max": [int(kns.min()), round(float(kns.mean()), 1), int(kns.max())],
                      "estimation_ok": f"{int(sum(x.estimation_ok for x in lst))}/{n}"}
            if cpu_legs and oparams is not None:
                from oracle import oracle
                hs = sorted({0, 5, n - 1})   # teacher-forced below (5: a sequence with the scene cut)
                hframes = [f for sc in scenes for f in sc]
                logh = r3.ehs[0].read_nav_log_array(0, Wm + K)
                r3.close()
                r3 = None
                wide = None
                if oracle.available("ref"):
                    # free-running, every 32nd sequence (+ the scene-cut one): how many leave the reference, where, and whether
                    # on a frame the reference itself leaves undecided
                    hw = sorted(set(range(0, n, max(1, n // 32))) | {5, n - 1})
                    wide, wide_trajs = wide_parity(logh, hw, lambda s_: [hidx(k, s_) for k in range(Wm + K)], np.stack(hframes),
                                                   ("euroc", w, h, FRAME_DT), Wm, max(1, _usable_cores() - 1))
                    # ... and every departed sequence (up to four more) joins the teacher-forced replay
                    hs = sorted(set(hs) | {d_["sequence"] for d_ in wide["departures"][:4]})
                gt = {s: _traj_of_log(logh[Wm:], s) for s in (hw if wide else hs)}
                if oracle.available("ref"):
                    # One pass of the reference per checked sequence serves three comparisons (oracle/teacher.py):
                    #  * free-running: the batch's own trajectory against the reference's (pose_rmse, as in the main leg);
                    #  * teacher-forced: a one-sequence context receives the reference's state before every frame — every
                    #    frame must agree inside the per-frame tolerance, knife-edge frames included;
                    #  * attribution: where the free-running trajectory leaves the reference, the frame must be one the
                    #    reference's own arithmetic leaves undecided (a KeyLine detected exactly on a half pixel whose
                    #    re-projection at X = 0 rounds either way, oracle.half_pixel_keylines) — the reference run with
                    #    another LAPACK or on another CPU model parts from itself on the same frames.
                    from oracle import teacher
                    ref_trajs, forced, depart = {}, {}, {}
                    for s in hs:
                        orc = oracle.Oracle("ref", oparams)
                        eh1 = edgehip.EdgeHip(params, nseq=1, nslots=3, device=local_rank)
                        tf = teacher.teacher_forced_replay(eh1, orc, lambda k, s=s: hframes[hidx(k, s)], Wm + K)
                        eh1.close()
                        orc.close()
                        ref_trajs[s] = {k: tf["ref"][Wm + k] for k in range(K)}
                        kef = [f["frame"] for f in tf["knife_edge_frames"]]
                        forced[int(s)] = {"frames": tf["frames"], "max_dV": tf["max_dV"], "max_dW": tf["max_dW"],
                                          "outside_tolerance": tf["outside_tolerance"], "knife_edge_frames": kef}
                        first = None
                        for k in range(1, Wm + K):
                            rv, rw = tf["ref"][k][2], tf["ref"][k][3]
                            if not (np.all(np.isfinite(rv)) and np.all(np.isfinite(rw))):
                                continue
                            tol = 1e-6 * (np.linalg.norm(rv) + np.linalg.norm(rw)) + 1e-9
                            if max(np.max(np.abs(logh[k, s]["V"] - rv)), np.max(np.abs(logh[k, s]["W"] - rw))) > tol:
                                first = k
                                break
                        depart[int(s)] = None if first is None else {
                            "frame": first, "knife_edge_frame": first in kef,
                            "keylines": next((f["keylines"] for f in tf["knife_edge_frames"] if f["frame"] == first), None)}
                    hetero["pose_rmse"] = pose_rmse(gt, wide_trajs if wide else ref_trajs, "reference")
                    hetero["free_running_parity"] = wide
                    hetero["teacher_forced"] = {
                        "what": "reference state injected before every frame (previous edge map with depths, velocity prior, pose, "
                                "threshold), one frame run, |dV|,|dW| <= 1e-6*step + 1e-9 and identical kn / klm_num / EstimationOK required",
                        "per_sequence": forced,
                        "frames_outside_tolerance": int(sum(len(v["outside_tolerance"]) for v in forced.values()))}
                    hetero["free_running_departures"] = {
                        "what": "first frame on which the batch's own trajectory differs from the reference's by more than the per-frame "
                                "tolerance; knife_edge_frame: the reference's own result for that frame hangs on the last bits of a depth "
                                "(DESIGN.md section 5; profiles/r03_knife_edge_*.txt)",
                        "per_sequence": depart}
                else:
                    hetero["pose_rmse"] = pose_rmse(gt, {s: _cpu_traj(oracle, oparams, lambda k, s=s: hframes[hidx(k, s)], Wm, K) for s in hs},
                                                    "restatement")
            if r3 is not None:
                r3.close()
        except Exception as e:
            hetero = dict(hetero or {"value": None})
            hetero["error"] = f"{type(e).__name__}: {e}"[:300]

    full = {
        "metric": ("frames/sec (DoG+extract+track+depth, ImuMode=2) 752x480 EuRoC" if args.imu else
                   "frames/sec (DoG+extract+track+depth) 752x480 EuRoC") if not tum else
                  "frames/sec (undistort+DoG+extract+track+depth) 640x480 TUM",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 scale-space / f64 tracker+EKF", "data": data_kind,
        "config": {"workload": ("full path (configs[2] at configs[1] size): 752x480 " + ("synthetic EuRoC-intrinsics " if data_kind == "synthetic" else
                                f"{data_kind} data set (sequences of the batch start at staggered frames of the mounted list) ") +
                                "sequences, GlobalConfig_EuRoC params, " + ("ImuMode=2 (the IMU branch of SecondThread batched on the device, "
                                "integrated IMU data synthesised from the camera motion)" if args.imu else "ImuMode=0")) if not tum else
                               ("BASELINE configs[3]: 640x480 synthetic TUM-intrinsics sequences taken as the distorted camera "
                                "image, GlobalConfig_desk.txt params, UseUndistort=1 with the EuRoC distortion (SURVEY 8d scene S3)"),
                   "input": in_note, "dataset": data_note,
                   "sequences_per_gpu": B * C, "contexts_per_gpu": C, "sequences_per_launch": B,
                   "stream_overlap": bool(args.overlap), "nav_gather": nav_gather, "nav_gather_info": nav_gather_info,
                   "frames_per_step": B * C * world, "keylines_per_frame": round(kn_mean, 1),
                   "keylines_per_frame_timed_mean": round(kn_timed, 1),
                   "tryvelrot_evals_per_frame": evals, "estimation_ok": f"{ok}/{B * C}",
                   "algorithmic_MB_per_frame": round(frame_bytes / 1e6, 2),
                   "whole_path_hbm_frac": round(frame_bytes * value / world / 1e9 / HBM_PEAK_GBS, 5)},
        "roofline": roof, "cpu_baseline": cpu, "pose_rmse": pose, "kernel_us_per_step": breakdown,
        "kernel_us_per_step_source": "HIP events of the 4 steps that follow the timed region (profiler on every kernel group; the timed "
                                     "region itself carries events on the dominant group only)",
        "roofline_kernels": roof_all,
        # no SCALE record of this repository exists (the driver's 8-GPU node has not been available): N > 1 is covered by the gloo
        # tests and a 2-rank dry run, not by a measured curve
        "scaling_measured": False,
        "traffic_source": ("live: " + LIVE_PMC["note"] + f" ({LIVE_PMC.get('seconds')} s, {LIVE_PMC.get('kernels')} kernels)") if LIVE_PMC["used"] else
                          (PMC_FILE + " (rocprofv3 --pmc passes of this command, committed; used only when stamped with the sources of "
                                      "the library that is running: " + pmc_stamp_note() + "); live passes: " + LIVE_PMC["note"]),
        "traffic_calibration": {"factors_true_over_reported": {k: round(v, 3) for k, v in calib.items()}, "source": calib_src},
    }
    full["launched_by"] = "bench.py itself (no launcher around the command)" if os.environ.get("BENCH_SELF_LAUNCHED") == "1" else \
                          ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "direct")
    if f32:
        full["metric"] = full["metric"].replace("(DoG+extract+track+depth", "(float tracker: DoG+extract+track+depth")
        full["dtype"] = "f32 scale-space / f32 tracker (Minimizer_RV<float>, TryVelRot<float>) / f64 matcher+EKF"
        full["config"]["workload"] += "; SECOND CONFIGURATION: the tracker's float instantiation (global_tracker.cpp:824, the reference's USE_NE10 build)"
        full["config"]["tracker"] = "Minimizer_RV<float>: 12 evaluations per frame in 9 launches (k_try_velrot2_f32 x3, k_try_velrot_f32 x6), LM step rounding JtJ / JtF / h / X to float"
    if stub:
        full["invalid_as_measurement"] = True
        full["data"] = "stub"
        full["config"]["workload"] = ("LAUNCHER / CONTROL-FLOW DRY RUN on tests/stub_device.py: no device work, `value` is not a measurement "
                                      "(self-launch of the ranks, process group, barriers, nav gather, max-over-ranks timing, one line)")
        full["dtype"] = "none (tests/stub_device.py: no device work)"
    if surface:
        full["host_surface"] = surface
    if world == 1 and args.extras and not args.no_extras and args.config == "full" and not args.imu:
        full["extras"] = other_configs(args)
        full["pcie_inclusive"] = pcie_inclusive(edgehip, params, frames, offs, B, K, max(Wm, 4), local_rank)
        if full["pcie_inclusive"].get("grey8", {}).get("value"):
            full["pcie_inclusive"]["grey8_over_resident"] = round(full["pcie_inclusive"]["grey8"]["value"] / value, 3)
        full["pcie_inclusive"]["note"] = ("`value` (the headline) has its inputs resident in HBM before the timed region, as the bench "
                                          "contract asks; these are the same replay with every frame crossing the link inside it")
    if sweep:
        full["batch_sweep"] = sweep
        full["single_sequence_ms_per_frame"] = sweep[0]["ms_per_step"]
        full["single_sequence_note"] = ("rate of back-to-back frames of one sequence; for batches below the one-kernel stage A's threshold the "
                                        "library runs the next frame's detection beside this frame's tracking and mapping (EDGEHIP_OVERLAP "
                                        "default), so one frame's way through the path is ~15 % longer than this figure")
    if hetero:
        full["heterogeneous"] = hetero
    emit(full)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
