"""GPU: batch groups behind the rebvo::REBVO plugin surface (rebvo_amd/host/src/batch_group.cpp).

N objects whose configs name the same &GPU BatchGroup share one edgehip context of N sequences; every object is fed through
requestCustomCamBuffer / releaseCustomCamBuffer and read through its own output callback and getNav().  Checked here:

* every object of a group of 8 against the reference oracle on ITS frames (same bounds as tests/test_host_gpu.py), and against
  the same 8 sequences run as one ctypes batch through edgehip_process_frame: the records must be bit-identical (the group is the
  batch, nothing else);
* a group of one (an object without a BatchGroup: the same pipelined engine) against the ctypes batch too;
* a member that leaves (CleanUp) while the others carry on; members that do not fit together are refused at Init()."""
import json
import os
import subprocess

import numpy as np
import pytest

from rebvo_amd import edgehip, synth
from tests.helpers import require_ref, write_global_config

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rebvo_amd", "lib", "surface_replay")
W, H = 376, 240
T0, DT = 1.0, 0.05


def tri(k, n):
    p = 2 * (n - 1)
    k %= p
    return k if k < n else p - k


def _run(tmp_path, frames, n_obj, n_fr, extra, tag="run", gpu=None, env=None, params=None):
    if not os.path.exists(EXE):
        pytest.fail("surface_replay not built — a broken snapshot: run __graft_entry__.build()")
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    cfg = tmp_path / "cfg"
    write_global_config(cfg, params if params is not None else edgehip.euroc_params(W, H), gpu=gpu)
    prefix = tmp_path / tag
    r = subprocess.run([EXE, str(cfg), str(tmp_path / "frames.rgb24"), str(len(frames)), str(n_obj), str(n_fr), str(T0), str(DT),
                        "--dump", str(prefix)] + extra, capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    js = json.loads(r.stdout.strip().splitlines()[-1])
    return js, [np.loadtxt(f"{prefix}.{i}.txt", ndmin=2) for i in range(n_obj)], r.stdout


def _ctypes_batch(frames, n_obj, n_fr, tint=None, params=None, tracker_bits=64):
    """The same sequences as one batch through the C-ABI: per step the nav records, and the old slot's KeyLines after the step.
    tint = (object, frame): that frame's first byte flipped, as surface_replay --tint does."""
    eh = edgehip.EdgeHip(params if params is not None else edgehip.euroc_params(W, H), nseq=n_obj, nslots=3, device=0)
    if tracker_bits != 64:
        eh.set_tracker_precision(tracker_bits)
    navs, kls = [], []
    for k in range(n_fr):
        batch = np.stack([frames[tri(k + i, len(frames))] for i in range(n_obj)])
        if tint is not None and tint[1] == k:
            batch[tint[0]].reshape(-1)[0] ^= 0x80
        eh.upload_rgb(eh.next_slot(), batch)
        eh.process_frame(np.full(n_obj, T0 + DT * k))
        navs.append([(np.array(n.Pos[:]), np.array(n.PoseLie[:]), np.array(n.Vel[:]), n.kn, n.klm_num, n.estimation_ok) for n in eh.read_nav()])
        if k:
            so = (eh.cur_slot() + 2) % 3
            kls.append([eh.download_keylines(i, so, want_mask=False)[0] for i in range(n_obj)])
    eh.close()
    return navs, kls


def _check_against_batch(rows, navs, kls, i, n_deliv):
    assert len(rows) == n_deliv
    for j in range(n_deliv):          # frame j is delivered once frame j + 1 has been tracked
        row = rows[j]
        assert int(row[0]) == j and abs(row[1] - (T0 + DT * j)) < 1e-12
        kl = kls[j][i]                # slot of frame j after frame j + 1 went over it
        assert int(row[2]) == len(kl)
        if j > 0:
            pos, lie, vel, kn, klm, ok = navs[j][i]
            assert int(row[3]) == klm and int(row[4]) == ok
            assert np.array_equal(row[5:8], pos) and np.array_equal(row[8:11], lie) and np.array_equal(row[11:14], vel)
        # the callback's loop adds KeyLine by KeyLine: the same order as a running sum
        assert row[14] == (np.cumsum(kl["rho"])[-1] if len(kl) else 0.0)
        assert row[15] == (np.cumsum(kl["s_rho"])[-1] if len(kl) else 0.0)


def test_group_of_eight_objects_equals_the_batch_and_follows_the_reference(tmp_path):
    oracle = require_ref()
    n_obj, n_fr, pool = 8, 7, 6
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool)]
    js, dumps, out = _run(tmp_path, frames, n_obj, n_fr, ["--group", "rig", "--threads", "3"])
    assert js["objects"] == n_obj and js["group"] == "rig" and js["callbacks"] == n_obj * (n_fr - 1), out
    navs, kls = _ctypes_batch(frames, n_obj, n_fr)
    for i in range(n_obj):
        _check_against_batch(dumps[i], navs, kls, i, n_fr - 1)
    # ... and every object against the reference on its own frames
    for i in (0, 3, 7):
        orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
        path = 0.0
        rn = []
        for k in range(n_fr):
            _, nav = orc.process_frame(frames[tri(k + i, pool)], T0 + DT * k)
            rn.append(nav)
            if k == 0:
                continue
            j = k - 1
            row = dumps[i][j]
            kl = orc.keylines(j % 8)
            assert int(row[2]) == len(kl)
            if j > 0:
                assert int(row[4]) == rn[j].estimation_ok
                path += np.linalg.norm(rn[j].V[:])
                assert np.allclose(row[5:8], rn[j].Pos[:], atol=1e-6 * path + 1e-9)
                assert np.allclose(row[8:11], rn[j].PoseLie[:], atol=1e-7)
                assert np.allclose(row[11:14], rn[j].Vel[:], rtol=1e-5, atol=1e-9)
            assert abs(row[14] - kl["rho"].sum()) <= 1e-6 * abs(kl["rho"].sum()) + 1e-9
            assert abs(row[15] - kl["s_rho"].sum()) <= 1e-6 * abs(kl["s_rho"].sum()) + 1e-9
        orc.close()


def test_objects_without_a_group_run_the_same_engine_one_context_each(tmp_path):
    n_obj, n_fr, pool = 2, 6, 6
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=5)]
    js, dumps, out = _run(tmp_path, frames, n_obj, n_fr, ["--threads", "2"])
    assert js["group"] is None and js["callbacks"] == n_obj * (n_fr - 1), out
    navs, kls = _ctypes_batch(frames, n_obj, n_fr)      # sequences of a batch are independent: the batch is the yardstick here too
    for i in range(n_obj):
        _check_against_batch(dumps[i], navs, kls, i, n_fr - 1)


def test_a_member_that_leaves_does_not_stop_the_group(tmp_path):
    n_obj, n_fr, pool, leave_at = 3, 7, 6, 3
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=9)]
    js, dumps, out = _run(tmp_path, frames, n_obj, n_fr, ["--group", "trio", "--leave", f"1:{leave_at}"])
    navs, kls = _ctypes_batch(frames, n_obj, n_fr)
    for i in (0, 2):
        _check_against_batch(dumps[i], navs, kls, i, n_fr - 1)
    # the member that left: what it was given before is right, and no more than its frames minus the one never delivered
    assert len(dumps[1]) <= leave_at - 1
    _check_against_batch(dumps[1], navs, kls, 1, len(dumps[1]))


def test_members_that_do_not_fit_are_refused(tmp_path):
    """Two objects name one group but differ in a tracker parameter / in BatchSize: the second Init() fails with the reason, the
    first one is left waiting for a partner and shuts down cleanly."""
    import ctypes as C
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, 2)]
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    write_global_config(tmp_path / "cfg", edgehip.euroc_params(W, H))
    # surface_replay builds every member from the same parameters, so a misfit is provoked with BatchSize: 1 object, group of... 1 is
    # fine; ask for a second run in which the group is already complete
    r = subprocess.run([EXE, str(tmp_path / "cfg"), str(tmp_path / "frames.rgb24"), "2", "1", "3", "0", "0.05", "--group", "solo"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr          # a named group of one works like any other
    lib = C.CDLL(os.path.join(ROOT, "rebvo_amd", "lib", "librebvohost.so"))
    assert hasattr(lib, "rebvo_group_selftest")
    lib.rebvo_group_selftest.restype = C.c_int
    lib.rebvo_group_selftest.argtypes = [C.c_char_p]
    assert lib.rebvo_group_selftest(str(tmp_path / "cfg").encode()) == 0


def test_frame_by_frame_mode_and_a_snapshot(tmp_path):
    """toggleFrameByFrame / advanceFrameByFrame (rebvo.h:481-488, rebvo_first_t.cpp:154-159): an object in step mode takes a frame only
    after the application has said so — here before every frame, so the run is the same run; TakeSnapshot (rebvo.h:459,
    rebvo_third_t.cpp:335-343): the next delivered frame's image lands in Snap0.ppm, the reference's P6 file."""
    n_obj, n_fr, pool = 2, 6, 6
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=13)]
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    cfg = tmp_path / "cfg"
    write_global_config(cfg, edgehip.euroc_params(W, H))
    r = subprocess.run([EXE, str(cfg), str(tmp_path / "frames.rgb24"), str(pool), str(n_obj), str(n_fr), str(T0), str(DT), "--group", "steps",
                        "--dump", str(tmp_path / "run"), "--step-mode", "--snapshot-at", "2"], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert r.stdout.count("Advancing frame...") in (n_fr, n_fr + 1)      # one per frame (+ one taken while waiting for a frame that never comes)
    dumps = [np.loadtxt(f"{tmp_path}/run.{i}.txt", ndmin=2) for i in range(n_obj)]
    navs, kls = _ctypes_batch(frames, n_obj, n_fr)
    for i in range(n_obj):
        _check_against_batch(dumps[i], navs, kls, i, n_fr - 1)
    snap = (tmp_path / "Snap0.ppm").read_bytes()
    head = f"P6\n{W} {H} 255\n".encode()
    assert snap.startswith(head) and len(snap) == len(head) + W * H * 3
    img = np.frombuffer(snap[len(head):], np.uint8).reshape(H, W, 3)
    # the request came before frame 2 was submitted: the output thread honours it with the frame it delivers next — one of object 0's
    # frames up to 2 (delivery runs a frame or two behind submission)
    assert any(np.array_equal(img, frames[tri(k + 0, pool)]) for k in range(0, 3))


def test_batched_keyline_download_equals_the_single_one():
    """edgehip_download_keylines_batch (what a group's callbacks are served from): one packing kernel for several sequences — record
    for record, byte for byte, what edgehip_download_keylines returns, on the slot the frame driver rotated out of place and on the newest."""
    n_obj, pool = 5, 6
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=21)]
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=n_obj, nslots=3, device=0)
    for k in range(4):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[tri(k + i, pool)] for i in range(n_obj)]))
        eh.process_frame(np.full(n_obj, T0 + DT * k))
    for slot in ((eh.cur_slot() + 2) % 3, eh.cur_slot()):
        seqs = [4, 0, 2]
        for registered in ((), (0, 2)):      # (0, 2): those two destinations page-locked in place, the copy lands in them directly
            got = eh.download_keylines_batch(slot, seqs, registered=registered)
            for s_, kl in zip(seqs, got):
                ref = eh.download_keylines(s_, slot, want_mask=False)[0]
                assert len(kl) == len(ref) > 1000 and kl.tobytes() == ref.tobytes()
    import ctypes as C
    assert eh.lib.edgehip_unregister_host(C.c_void_p(12345)) != 0      # not a registered range
    assert eh.lib.edgehip_register_host(None, C.c_size_t(0)) != 0
    eh.close()


def test_in_stream_keyline_export_equals_the_synchronising_download_with_frames_in_flight():
    """edgehip_export_keylines / _fetch / _wait (what a group's callbacks are served from since round 6): the old slot's lists packed
    behind frame k without a synchronisation, fetched and waited for only after frames k+1 and k+2 have been enqueued — k+2 detects
    into the very slot the lists came from.  Byte for byte what edgehip_download_keylines returned for that slot right behind frame k
    (a second context run in lock-step provides that)."""
    import ctypes as C
    n_obj, pool, n_fr = 5, 6, 9
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=23)]
    p = edgehip.euroc_params(W, H)
    eh, ref = edgehip.EdgeHip(p, nseq=n_obj, nslots=3, device=0), edgehip.EdgeHip(p, nseq=n_obj, nslots=3, device=0)
    seqs = [4, 0, 2]
    want, tickets, got = {}, {}, {}

    def collect(k):
        kns = [len(want[k][j]) for j in range(len(seqs))]
        f = eh.export_fetch(tickets.pop(k), kns, registered=(k % 2 == 0))      # page-locked and pageable destinations alike
        got[k] = eh.export_wait(f)
    for k in range(n_fr):
        batch = np.stack([frames[tri(k + i, pool)] for i in range(n_obj)])
        for e in (eh, ref):
            e.upload_rgb(e.next_slot(), batch)
            e.process_frame(np.full(n_obj, T0 + DT * k))
        if k >= 1:
            tickets[k] = eh.export_keylines(seqs)                               # in-stream, no synchronisation
            want[k] = [ref.download_keylines(s_, (ref.cur_slot() + 2) % 3, want_mask=False)[0] for s_ in seqs]
        if k >= 3:
            collect(k - 2)                                                      # two frames later: its slot has been detected into again
    for k in list(tickets):
        collect(k)
    assert sorted(got) == list(range(1, n_fr))
    for k in got:
        for a, b in zip(got[k], want[k]):
            assert len(a) == len(b) > 1000 and a.tobytes() == b.tobytes(), k
    # the ticket discipline: a fifth outstanding ticket is refused, an unknown one too, a dropped one frees its entry
    held = [eh.export_keylines(seqs) for _ in range(4)]
    t = C.c_int(0)
    arr = np.array(seqs, np.int32)
    assert eh.lib.edgehip_export_keylines(eh.ctx, 3, arr.ctypes.data_as(C.c_void_p), C.byref(t)) != 0
    assert eh.lib.edgehip_export_wait(eh.ctx, 123456) != 0
    for h in held:
        eh.export_drop(h)
    eh.export_drop(eh.export_keylines(seqs))
    fresh = edgehip.EdgeHip(p, nseq=2, nslots=3, device=0)
    assert fresh.lib.edgehip_export_keylines(fresh.ctx, 1, arr.ctypes.data_as(C.c_void_p), C.byref(t)) != 0      # no frame pair yet
    for e in (eh, ref, fresh):
        e.close()


def test_a_snapshot_without_a_callback_holds_a_real_frame(tmp_path):
    """TakeSnapshot() on an object that has NO output callback: the group engine keeps a frame's image for the output
    thread only when a callback or a snapshot request is pending when the frame is launched, and delivery runs behind launch — the
    snapshot must wait for a frame launched after the request and hold that frame, not the stale image of one launched before it."""
    n_obj, n_fr, pool = 2, 7, 6
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=13)]
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    cfg = tmp_path / "cfg"
    write_global_config(cfg, edgehip.euroc_params(W, H))
    r = subprocess.run([EXE, str(cfg), str(tmp_path / "frames.rgb24"), str(pool), str(n_obj), str(n_fr), str(T0), str(DT), "--group", "snap",
                        "--snapshot-at", "4"], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["callbacks"] == 0
    snap = (tmp_path / "Snap0.ppm").read_bytes()
    head = f"P6\n{W} {H} 255\n".encode()
    assert snap.startswith(head) and len(snap) == len(head) + W * H * 3
    img = np.frombuffer(snap[len(head):], np.uint8).reshape(H, W, 3)
    assert img.any()
    # one of the object's own frames that was LAUNCHED after the request — the application fills its four-entry camera ring before the
    # group's thread has launched anything, so that can be any frame from 0 on — never the zero image of a PipeBuffer nobody filled
    # (what the unlatched version saved)
    assert any(np.array_equal(img, frames[tri(k, pool)]) for k in range(0, n_fr))


def test_a_member_whose_ring_runs_ahead_after_a_dropped_frame(tmp_path):
    """The soft-FPS gate (rebvo_first_t.cpp:146, 172-177) per member: one object submits a frame too many (stamped like the one
    before it) — dropped, and from then on that object's camera ring stands one entry ahead of the others', so its frames no longer
    lie next to theirs in the group's page-locked ring and go up in a copy of their own.  Results: as if nothing had happened."""
    n_obj, n_fr, pool = 4, 7, 6
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=17)]
    js, dumps, out = _run(tmp_path, frames, n_obj, n_fr, ["--group", "drop", "--dup", "2:3"])
    navs, kls = _ctypes_batch(frames, n_obj, n_fr)
    for i in range(n_obj):
        rows = dumps[i].copy()
        if i == 2:      # the dropped frame took a camera sequence number (p_id counts frames grabbed, rebvo_first_t.cpp:89)
            assert [int(r[0]) for r in rows] == [0, 1, 2, 4, 5, 6][:len(rows)]
            rows[:, 0] = np.arange(len(rows))
        _check_against_batch(rows, navs, kls, i, n_fr - 1)


def test_mono_frames_cross_pcie_as_8_bit_planes_and_nothing_else_changes(tmp_path):
    """&GPU MonoUpload (default on; rebvo_amd/host/src/mono_pack.cpp): a step whose frames all have R = G = B goes up as 8-bit planes
    (edgehip_upload_grey8_pinned), any other step as RGB24 — per step, so one coloured pixel in one member's frame sends that step's
    frames up in full.  The records and KeyLines are those of the RGB24 batch either way, bit for bit."""
    import re
    n_obj, n_fr, pool = 4, 7, 6
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=29)]
    assert all(np.array_equal(f[..., 0], f[..., 1]) and np.array_equal(f[..., 1], f[..., 2]) for f in frames)
    timing = {"REBVO_GROUP_TIMING": "1"}
    planes = lambda out: int(re.search(r"(\d+) steps \((\d+) as 8-bit planes\)", out).group(2))
    navs, kls = _ctypes_batch(frames, n_obj, n_fr)
    for tag, gpu, want in (("mono", dict(group="m", size=n_obj), n_fr), ("rgb", dict(group="m", size=n_obj, mono=0), 0)):
        js, dumps, out = _run(tmp_path, frames, n_obj, n_fr, [], tag=tag, gpu=gpu, env=timing)
        assert planes(out) == want, out[-600:]
        for i in range(n_obj):
            _check_against_batch(dumps[i], navs, kls, i, n_fr - 1)
    navs, kls = _ctypes_batch(frames, n_obj, n_fr, tint=(1, 3))
    js, dumps, out = _run(tmp_path, frames, n_obj, n_fr, ["--tint", "1:3"], tag="tint", gpu=dict(group="m", size=n_obj), env=timing)
    assert planes(out) == n_fr - 1, out[-600:]
    for i in range(n_obj):
        _check_against_batch(dumps[i], navs, kls, i, n_fr - 1)


def test_without_callbacks_two_steps_in_flight_final_poses_equal_the_batch(tmp_path):
    """The configuration bench.py's host_surface times: no callback, so two steps in flight, the next step's copy enqueued ahead, mono
    frames as 8-bit planes — 32 objects over 60 frames, four producer threads.  What getNav() shows at the end must be the ctypes
    batch's last record, digit for digit; and again with RGB24 uploads (MonoUpload=0)."""
    import re
    n_obj, n_fr, pool = 32, 60, 8
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=31)]
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=n_obj, nslots=3, device=0)
    for k in range(n_fr):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[tri(k + i, pool)] for i in range(n_obj)]))
        eh.process_frame(np.full(n_obj, T0 + DT * k))
    want = [np.array(n.Pos[:]) for n in eh.read_nav()]
    oks = sum(n.estimation_ok for n in eh.read_nav())
    eh.close()
    assert oks == n_obj
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    for tag, gpu, planes in (("mono", dict(group="q", size=n_obj), n_fr), ("rgb", dict(group="q", size=n_obj, mono=0), 0)):
        cfg = tmp_path / f"cfg_{tag}"
        write_global_config(cfg, edgehip.euroc_params(W, H), gpu=gpu)
        r = subprocess.run([EXE, str(cfg), str(tmp_path / "frames.rgb24"), str(pool), str(n_obj), str(n_fr), str(T0), str(DT), "--threads", "4"],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, REBVO_GROUP_TIMING="1"))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        js = json.loads(r.stdout.strip().splitlines()[-1])
        assert js["callbacks"] == 0 and js["objects"] == n_obj
        m = re.search(r"(\d+) steps \((\d+) as 8-bit planes\), look-ahead copies (\d+)", r.stdout)
        assert m and int(m.group(1)) == n_fr and int(m.group(2)) == planes, r.stdout[-800:]
        got = {int(a): np.array([float(x), float(y), float(z)]) for a, x, y, z in
               re.findall(r"object (\d+) final Pos = (\S+) (\S+) (\S+)", r.stdout)}
        assert len(got) == n_obj
        for i in range(n_obj):
            assert np.array_equal(got[i], want[i]), (tag, i, got[i], want[i])


def test_tum_configuration_with_undistortion_through_the_surface(tmp_path):
    """BASELINE configs[3] behind rebvo::REBVO: four objects in one group, 640x480, GlobalConfig_desk.txt parameters,
    UseUndistort=1 (image_undistort::undistort<true>, include/VideoLib/image_undistort.h:66-122, in front of ConvertRGB2BW — the
    undistortion pre-pass of the batch's stage A).  The frames are a mono camera's, so the steps cross PCIe as 8-bit planes — except the
    step in which one member's frame carries a coloured pixel, which goes up as RGB24: both upload formats meet the undistortion.
    Every object bit-identical to the same four sequences as one ctypes batch, and within the usual bounds of the reference (its own
    undistorter, its own frames, the tinted one included)."""
    import re
    oracle = require_ref()
    w, h, n_obj, n_fr, pool = 640, 480, 4, 7, 6
    p = edgehip.tum_params(w, h, use_undistort=1)
    intr = dict(fx=float(p.zfx), fy=float(p.zfy), cx=float(p.ppx), cy=float(p.ppy))
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, pool, seed=37, **intr)]
    tint = (1, 3)
    js, dumps, out = _run(tmp_path, frames, n_obj, n_fr, ["--tint", "%d:%d" % tint, "--threads", "2"], gpu=dict(group="tum", size=n_obj),
                          env={"REBVO_GROUP_TIMING": "1"}, params=p)
    m = re.search(r"(\d+) steps \((\d+) as 8-bit planes\)", out)
    assert m and int(m.group(1)) == n_fr and int(m.group(2)) == n_fr - 1, out[-600:]
    navs, kls = _ctypes_batch(frames, n_obj, n_fr, tint=tint, params=p)
    for i in range(n_obj):
        _check_against_batch(dumps[i], navs, kls, i, n_fr - 1)
    for i in (0, 1, 3):
        orc = oracle.Oracle("ref", oracle.tum_params(w, h, use_undistort=1))
        path, rn = 0.0, []
        for k in range(n_fr):
            f = frames[tri(k + i, pool)]
            if (i, k) == tint:
                f = f.copy()
                f.reshape(-1)[0] ^= 0x80
            _, nav = orc.process_frame(f, T0 + DT * k)
            rn.append(nav)
            if k == 0:
                continue
            j = k - 1
            row = dumps[i][j]
            assert int(row[2]) == len(orc.keylines(j % 8)), (i, j)
            if j > 0:
                assert int(row[4]) == rn[j].estimation_ok and int(row[3]) == rn[j].klm_num, (i, j)
                path += np.linalg.norm(rn[j].V[:])
                assert np.allclose(row[5:8], rn[j].Pos[:], atol=1e-6 * path + 1e-9), (i, j)
                assert np.allclose(row[8:11], rn[j].PoseLie[:], atol=1e-7), (i, j)
        orc.close()


def test_the_float_tracker_behind_the_surface(tmp_path):
    """&GPU TrackerPrecision=32 (the run-time form of the reference's USE_NE10 switch, rebvo_second_t.cpp:339-346): three objects in a group
    run Minimizer_RV<float> on the device; every object's callback rows are those of the ctypes batch with edgehip_set_tracker_precision(32),
    bit for bit — and not those of the fp64 tracker.  (Parity of the float tracker itself: tests/test_tracker_f32_gpu.py.)"""
    n_obj, n_fr, pool = 3, 7, 6
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=41)]
    js, dumps, out = _run(tmp_path, frames, n_obj, n_fr, [], gpu=dict(group="f32", size=n_obj, tracker_precision=32))
    navs, kls = _ctypes_batch(frames, n_obj, n_fr, tracker_bits=32)
    for i in range(n_obj):
        _check_against_batch(dumps[i], navs, kls, i, n_fr - 1)
    navs64, _ = _ctypes_batch(frames, n_obj, n_fr)
    assert any(not np.array_equal(dumps[0][j][5:8], navs64[j][0][0]) for j in range(1, n_fr - 1))
