"""GPU: stereo depth (REBVO/StereoAvaiable) — stage A of the pair image with the pair
camera's own intrinsics, directed_matching_stereo (+ search_match_stereo, getDepthFromStereo), fuseStereoDepth and
directed_matching in stereo mode — against the reference's own classes.

A main-camera sequence gives KeyLines with converged depth; the pair image is the last frame's scene rendered from the
EuRoC-like baseline hard-coded in rebvo_second_t.cpp:466-470, through slightly different intrinsics.
Match ids and counts: exact.  stereo_rho / stereo_s_rho and the fused depth: the same fp64 expressions -> 1e-12."""
import copy
import os
import subprocess

import numpy as np
import pytest

from rebvo_amd import edgehip, synth
from tests.helpers import to_edgehip_kl, write_global_config

pytestmark = pytest.mark.gpu
W, H, NF = 376, 240, 6
R_PAIR = np.array([[0.999997256477450, 0.002312067192420, 0.000376008102351],
                   [-0.002317135723285, 0.999898048506528, 0.014089835846697],
                   [-0.000343393120589, -0.014090668452670, 0.999900662638179]])
T_PAIR = np.array([-0.110073808127139, 0.000399121547014, -0.000853702503351])


def make_data(all_pairs=False, nf=NF):
    p = edgehip.euroc_params(W, H)
    scene = synth.BillboardScene(W, H, p.zfx, p.zfy, p.ppx, p.ppy, seed=11, ss=2)
    # the pair camera: X_pair = R_PAIR X_cam + T_PAIR, focal length / principal point of its own
    pair_cam = dict(ppx=p.ppx + 6.4, ppy=p.ppy + 3.4, zfx=p.zfx - 0.53, zfy=p.zfy - 0.58)
    s2 = copy.copy(scene)
    s2.fx, s2.fy, s2.cx, s2.cy = pair_cam["zfx"], pair_cam["zfy"], pair_cam["ppx"], pair_cam["ppy"]
    tw = synth.smooth_trajectory(nf, 13)
    R, t = np.eye(3), np.zeros(3)
    frames, pairs = [], []
    for k in range(nf):
        frames.append(np.repeat(scene.render(R, t)[:, :, None], 3, axis=2).copy())
        if all_pairs or k == nf - 1:
            pairs.append(np.repeat(s2.render(R_PAIR @ R, R_PAIR @ t + T_PAIR)[:, :, None], 3, axis=2).copy())
        dR = synth._so3_exp(tw[k, 3:])
        R, t = dR @ R, dR @ t + tw[k, :3]
    return p, frames, (pairs if all_pairs else pairs[-1]), pair_cam


def run_reference(p_unused, frames, pair, pair_cam):
    from oracle import oracle
    orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
    for k, f in enumerate(frames):
        orc.process_frame(f, 0.05 * k)
    s = orc.cur_slot()
    ps = (s + 3) % 8
    orc.set_slot_cam(ps, pair_cam["ppx"], pair_cam["ppy"], pair_cam["zfx"], pair_cam["zfy"])
    return orc, s, ps


def test_stereo_matches_reference():
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    p, frames, pair, pair_cam = make_data()
    orc, s, ps = run_reference(p, frames, pair, pair_cam)

    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H, stereo_available=1), nseq=2, nslots=3)
    # ---- stage A of the pair image through the pair camera: same detector state on both sides ----
    tresh, lkn = oracle.euroc_params(W, H).detector_thresh, 0
    o2 = oracle.Oracle("ref", oracle.euroc_params(W, H))
    o2.set_slot_cam(2, pair_cam["ppx"], pair_cam["ppy"], pair_cam["zfx"], pair_cam["zfy"])
    for k, f in enumerate(frames):
        tresh, lkn = o2.stage_a(k % 2, f, tresh, lkn)[-2:]
        eh.upload_rgb(k % 2, np.stack([f, f]))
        eh.stage_a(k % 2)
    eh.set_slot_camera(2, pair_cam["ppx"], pair_cam["ppy"], pair_cam["zfx"], pair_cam["zfy"])
    o2.stage_a(2, pair, tresh, lkn)
    eh.upload_rgb(2, np.stack([pair, pair]))
    eh.stage_a(2)
    kg, mg = eh.download_keylines(1, 2)
    kr = o2.keylines(2)
    assert len(kg) == len(kr) > 3000
    for f in ("p_inx", "m_m", "u_m", "n_m", "c_p", "p_m", "rho", "s_rho", "p_id", "n_id", "stereo_m_id", "stereo_rho", "stereo_s_rho"):
        assert np.array_equal(kg[f], kr[f]), f
    assert np.array_equal(mg, o2.mask(2).reshape(H, W))
    assert not np.array_equal(kr["p_m"], kr["c_p"] - np.array([p.ppx, p.ppy], np.float32))   # the pair's own principal point

    # ---- directed_matching_stereo on the tracked edge map of the main camera ----
    # pair edge map for the main oracle: copy the one just detected (KeyLines + mask) into its pair slot
    orc.set_keylines(ps, kr, o2.mask(2), o2.retuned(2))
    k_main = orc.keylines(s).copy()
    assert (k_main["s_rho"] < 10).sum() > 1000                         # depth has started to converge
    for seq in range(2):
        eh.upload_keylines(seq, 0, to_edgehip_kl(k_main), orc.mask(s), orc.retuned(s))
        eh.upload_keylines(seq, 1, to_edgehip_kl(kr), o2.mask(2), o2.retuned(2))
    eh.set_slot_camera(1, pair_cam["ppx"], pair_cam["ppy"], pair_cam["zfx"], pair_cam["zfy"])
    args = (T_PAIR, R_PAIR, p.match_thresh_module, p.match_thresh_angle, 100.0, p.loc_unc_match, p.reshape_q_abs, p.reshape_q_rel, p.loc_unc)
    n_ref = orc.directed_matching_stereo(s, ps, *args)
    n_gpu = eh.directed_matching_stereo(0, 1, *args)
    assert list(n_gpu) == [n_ref, n_ref] and n_ref > 300
    kg, _ = eh.download_keylines(1, 0, want_mask=False)
    kr1 = orc.keylines(s).copy()
    assert np.array_equal(kg["stereo_m_id"], kr1["stereo_m_id"])
    m = kr1["stereo_m_id"] >= 0
    assert np.allclose(kg["stereo_rho"], kr1["stereo_rho"], rtol=1e-12, atol=0)
    assert np.allclose(kg["stereo_s_rho"], kr1["stereo_s_rho"], rtol=1e-12, atol=0)
    assert (kr1["stereo_s_rho"][m] < 5).sum() > 100                    # informative depths came out of it
    # ambiguous / rejected candidates leave the initial values in place
    assert np.all(kr1["stereo_rho"][~m & (kr1["stereo_s_rho"] == 20.0)] == 1.0)

    # ---- fuseStereoDepth ----
    orc.fuse_stereo_depth(s)
    eh.fuse_stereo_depth(0)
    kg, _ = eh.download_keylines(0, 0, want_mask=False)
    kr2 = orc.keylines(s)
    for f in ("rho0", "s_rho0"):
        assert np.array_equal(kg[f], kr2[f]), f
    assert np.allclose(kg["rho"], kr2["rho"], rtol=1e-13, atol=0) and np.allclose(kg["s_rho"], kr2["s_rho"], rtol=1e-13, atol=0)
    assert np.any(kr2["rho"][m] != kr2["rho0"][m])

    # ---- without stereo_available the entry points refuse ----
    eh0 = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=1, nslots=2)
    with pytest.raises(RuntimeError):
        eh0.fuse_stereo_depth(0)



def test_directed_matching_stereo_mode():
    """directed_matching with StereoAvaiable: a match clones rho0 / s_rho0 and leaves rho_nr alone."""
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    p, frames, pair, pair_cam = make_data()
    orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
    for k, f in enumerate(frames[:-1]):
        orc.process_frame(f, 0.05 * k)
    so = orc.cur_slot()
    sn = (so + 1) % 8
    # detect the last frame without tracking it, then run the tracker stages by hand
    _, nav = orc.process_frame(frames[-1], 0.05 * (NF - 1))
    sn = orc.cur_slot()
    so = (sn + 7) % 8
    V, RVel = np.array(nav.V[:]), np.array(nav.P_V[:]).reshape(3, 3)
    R0 = np.array(nav.Rot[:]).reshape(3, 3)
    # make rho0 differ from rho on the old map so that the two modes are distinguishable
    ko = orc.keylines(so).copy()
    ko["rho0"] = ko["rho"] * 1.25
    ko["s_rho0"] = ko["s_rho"] * 0.5
    kn = orc.keylines(sn).copy()
    kn["m_id"] = -1
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H, stereo_available=1), nseq=1, nslots=2)
    orc.set_stereo_mode(1)
    orc.set_keylines(so, ko, orc.mask(so), orc.retuned(so))
    orc.set_keylines(sn, kn, orc.mask(sn), orc.retuned(sn))
    eh.upload_keylines(0, 0, to_edgehip_kl(ko), orc.mask(so), orc.retuned(so))
    eh.upload_keylines(0, 1, to_edgehip_kl(kn), orc.mask(sn), orc.retuned(sn))
    st = eh.get_state(0)
    st.V[:] = V
    st.P_V[:] = RVel.ravel()
    st.R[:] = R0.ravel()
    st.klm_num = 0
    st.kf_matchs = 0
    eh.set_state(0, st)
    n_ref, _ = orc.directed_matching(sn, so, V, RVel, R0, 1.0, 45.0, 40.0, 2.0)
    eh.directed_matching(1, 0)
    kg, _ = eh.download_keylines(0, 1, want_mask=False)
    kr = orc.keylines(sn)
    assert eh.get_state(0).klm_num == n_ref > 1000
    for f in ("m_id", "rho", "s_rho", "rho_nr", "s_rho_nr", "m_num"):
        assert np.array_equal(kg[f], kr[f]), f
    m = kr["m_id"] >= 0
    assert np.array_equal(kr["rho"][m], ko["rho0"][kr["m_id"][m]])


@pytest.mark.parametrize("w,h,nseq", [(376, 240, 2), (752, 480, 2), (376, 240, 200)])
def test_stereo_whole_frame_matches_reference(monkeypatch, w, h, nseq):
    """edgehip_process_frame with a stereo rig (stage A of the pair image, stereo-mode directed matching, stereo match,
    fuse, Kp = 1) against the reference's SecondThread order with StereoAvaiable, frame by frame; the sequences of the
    batch stay identical.  Also at BASELINE's 752x480, and with 200 sequences per launch, where both images of the pair go
    through the one-kernel stage A (k_stage_a_fused)."""
    import sys
    mod = sys.modules[__name__]
    monkeypatch.setattr(mod, "W", w)
    monkeypatch.setattr(mod, "H", h)
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    nf = 8
    p, frames, pairs, pc = make_data(all_pairs=True, nf=nf)
    orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
    orc.enable_stereo(pc["ppx"], pc["ppy"], pc["zfx"], pc["zfy"], T_PAIR, R_PAIR, 100.0)
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H, stereo_available=1), nseq=nseq, nslots=4)
    eh.set_slot_camera(3, pc["ppx"], pc["ppy"], pc["zfx"], pc["zfy"])
    eh.set_stereo_rig(3, T_PAIR, R_PAIR, 100.0)
    path = 0.0
    for k in range(nf):
        _, nr = orc.process_frame_stereo(frames[k], pairs[k], 0.05 * k)
        assert eh.next_slot() == k % 3                     # the ring leaves the pair slot alone
        eh.upload_rgb(eh.next_slot(), np.stack([frames[k]] * nseq))
        eh.upload_rgb(3, np.stack([pairs[k]] * nseq))
        eh.process_frame(0.05 * k)
        navs = eh.read_nav()
        nm = eh.get_stereo_matches()
        assert all(n_.V[:] == navs[0].V[:] and n_.kn == navs[0].kn for n_ in navs)
        for s, ng in list(enumerate(navs))[:2]:
            assert ng.kn == nr.kn and ng.tresh == nr.tresh, k   # the pair image went through the shared threshold state
            if k == 0:
                continue
            assert ng.estimation_ok == nr.estimation_ok == 1
            Vr, Wr = np.array(nr.V[:]), np.array(nr.W[:])
            step = np.linalg.norm(Vr) + np.linalg.norm(Wr)
            assert np.allclose(ng.V[:], Vr, rtol=0, atol=1e-6 * step + 1e-9), (k, ng.V[:], Vr)
            assert np.allclose(ng.W[:], Wr, rtol=0, atol=1e-6 * step + 1e-9)
            assert ng.Kp == nr.Kp == 1.0
            assert abs(int(nm[s]) - nr.pad0) <= max(2, nr.pad0 // 500), (k, nm, nr.pad0)
            assert abs(ng.klm_num - nr.klm_num) <= max(2, nr.klm_num // 1000)
        if k:
            path += np.linalg.norm(Vr)
            assert np.allclose(navs[0].Pos[:], nr.Pos[:], atol=1e-6 * path + 1e-9)
        assert navs[0].V[:] == navs[1].V[:]
    assert nr.pad0 > 1000
    kg, mask = eh.download_keylines(nseq - 1, eh.cur_slot())
    kr = orc.keylines(orc.cur_slot())
    assert np.array_equal(mask, orc.mask(orc.cur_slot()).reshape(H, W))
    same = (kg["m_id"] == kr["m_id"]) & (kg["stereo_m_id"] == kr["stereo_m_id"])
    assert same.mean() > 0.995
    for f in ("rho", "s_rho", "rho0", "s_rho0", "stereo_rho", "stereo_s_rho"):
        assert np.allclose(kg[f][same], kr[f][same], rtol=1e-5, atol=1e-7), f


def test_host_stereo_replay(tmp_path):
    """StereoAvaiable through the host library: an EuRoC-layout data set with cam0 and cam1 lists, the &Stereo
    intrinsics and the hard-coded rig, replayed by dataset_replay and compared with the reference's stereo sequence."""
    from oracle import oracle
    PIL = pytest.importorskip("PIL.Image")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "rebvo_amd", "lib", "dataset_replay")
    if not oracle.available("ref") or not os.path.exists(exe):
        pytest.fail("needs oracle/_ref and dataset_replay" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    nf = 8
    p, frames, pairs, pc = make_data(all_pairs=True, nf=nf)
    t_ns = [1403636579763555584 + 50_000_000 * k for k in range(nf)]
    dirs = {}
    for cam, imgs in (("cam0", frames), ("cam1", pairs)):
        d = tmp_path / "mav0" / cam
        (d / "data").mkdir(parents=True)
        with open(d / "data.csv", "w") as f:
            f.write("#timestamp [ns],filename\n")
            for k, fr in enumerate(imgs):
                PIL.fromarray(fr[:, :, 0], "L").save(d / "data" / f"{t_ns[k]}.png")
                f.write(f"{t_ns[k]},{t_ns[k]}.png\n")
        dirs[cam] = d
    cfg, dump = tmp_path / "cfg", tmp_path / "dump.txt"
    write_global_config(cfg, edgehip.euroc_params(W, H), camera_type=2,
                        dataset=(str(dirs["cam0"] / "data") + "/", str(dirs["cam0"] / "data.csv"), 1e-9),
                        stereo=dict(dir=str(dirs["cam1"] / "data") + "/", file=str(dirs["cam1"] / "data.csv"), **pc))
    r = subprocess.run([exe, str(cfg), str(dump)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rows = np.loadtxt(dump, ndmin=2)
    assert len(rows) == nf - 1
    orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
    orc.enable_stereo(pc["ppx"], pc["ppy"], pc["zfx"], pc["zfy"], T_PAIR, R_PAIR, 100.0)
    path, prev = 0.0, None
    for k in range(nf):
        t = float(np.float64(t_ns[k]) * 1e-9)
        _, nav = orc.process_frame_stereo(frames[k], pairs[k], t)
        if k == 0:
            prev = nav
            continue
        row = rows[k - 1]                       # frame k-1 is delivered after frame k was tracked
        kl = orc.keylines((k - 1) % 8)
        assert int(row[0]) == k - 1 and int(row[2]) == len(kl)
        if k - 1 > 0:
            path += np.linalg.norm(prev.V[:])
            assert np.allclose(row[5:8], prev.Pos[:], atol=1e-6 * path + 1e-9)
            assert row[27] == 1.0              # Kp = 1 with stereo
        assert abs(row[14] - kl["rho"].sum()) <= 1e-6 * abs(kl["rho"].sum()) + 1e-9
        prev = nav
    assert "Loaded 8 File names" in r.stdout


@pytest.mark.parametrize("mono_upload", [1, 0])
def test_stereo_objects_in_one_batch_group(tmp_path, mono_upload):
    """StereoAvaiable behind the plugin surface as ONE batch (`&GPU BatchGroup` admits stereo members, ImuMode 0): four rebvo::REBVO
    objects, custom cameras, each fed a pair frame (requestStereoCustomCamBuffer) and a main frame per instant, object i at its own
    phase of the pool.  The group's context has the pair slot behind its ring and the rig inside edgehip_process_frame
    (rebvo_second_t.cpp:465-486); pair frames cross in the group's second page-locked ring — as 8-bit planes (mono_upload = 1)
    or RGB24.  Checked: every object's callback rows bit-identical to the same four sequences run as one ctypes batch with the rig
    (records, KeyLine sums, stereo_match_num per frame although two steps are in flight), and object 0 / 3 against the reference's
    own SecondThread order with StereoAvaiable within the bounds of the tests above."""
    import json
    from oracle import oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "rebvo_amd", "lib", "surface_replay")
    if not oracle.available("ref") or not os.path.exists(exe):
        pytest.fail("needs oracle/_ref and surface_replay — a broken snapshot: run __graft_entry__.build()")
    B, n_fr, pool = 4, 8, 7
    p, frames, pairs, pc = make_data(all_pairs=True, nf=pool)
    t0, dt = 1.0, 0.05

    def tri(k, n):
        q = 2 * (n - 1)
        k %= q
        return k if k < n else q - k
    np.stack(frames).tofile(tmp_path / "frames.rgb24")
    np.stack(pairs).tofile(tmp_path / "pairs.rgb24")
    cfg = tmp_path / "cfg"
    write_global_config(cfg, edgehip.euroc_params(W, H), camera_type=3, dataset=("unused/", "unused.csv", 1.0),
                        stereo=dict(dir="unused/", file="unused.csv", **pc), gpu=dict(group="st4", size=B, mono=mono_upload))
    prefix = tmp_path / "run"
    r = subprocess.run([exe, str(cfg), str(tmp_path / "frames.rgb24"), str(pool), str(B), str(n_fr), repr(t0), repr(dt),
                        "--group", "st4", "--stereo", str(tmp_path / "pairs.rgb24"), "--dump", str(prefix), "--threads", "2"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, REBVO_GROUP_TIMING="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    js = json.loads(r.stdout.strip().splitlines()[-1])
    assert js["objects"] == B and js["callbacks"] == B * (n_fr - 1), r.stdout[-1500:]
    assert f"group 'st4': {n_fr} steps ({n_fr if mono_upload else 0} as 8-bit planes)" in r.stdout, r.stdout[-1500:]
    dumps = [np.loadtxt(f"{prefix}.{i}.txt", ndmin=2) for i in range(B)]

    # (a) the same four sequences as one ctypes batch with the rig
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H, stereo_available=1), nseq=B, nslots=4)
    eh.set_slot_camera(3, pc["ppx"], pc["ppy"], pc["zfx"], pc["zfy"])
    eh.set_stereo_rig(3, T_PAIR, R_PAIR, 100.0)
    eh.set_nav_log(8)
    navs, kls, nms = [], [], []
    for k in range(n_fr):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[tri(k + i, pool)] for i in range(B)]))
        eh.upload_rgb(3, np.stack([pairs[tri(k + i, pool)] for i in range(B)]))
        eh.process_frame(np.full(B, t0 + dt * k))
        navs.append(eh.read_nav())
        nms.append(eh.get_stereo_matches())
        assert np.array_equal(eh.read_stereo_matches_log(k, 1)[0], nms[-1] if k else np.zeros(B, np.int32))   # the log = the counter, frame by frame
        if k:
            kls.append([eh.download_keylines(i, (eh.cur_slot() + 2) % 3, want_mask=False)[0] for i in range(B)])
    eh.close()
    assert min(int(v) for v in nms[-1]) > 500
    arr = lambda v: np.array(v[:])
    for i in range(B):
        rows = dumps[i]
        assert len(rows) == n_fr - 1
        for j in range(n_fr - 1):                 # frame j is delivered once frame j + 1 has been tracked
            row, kl = rows[j], kls[j][i]
            assert int(row[0]) == j and abs(row[1] - (t0 + dt * j)) < 1e-12 and int(row[2]) == len(kl)
            assert row[14] == np.cumsum(kl["rho"])[-1] and row[15] == np.cumsum(kl["s_rho"])[-1], (i, j)
            if j == 0:
                assert int(row[-1]) == 0
                continue
            n = navs[j][i]
            assert int(row[3]) == n.klm_num and int(row[4]) == n.estimation_ok == 1, (i, j)
            assert np.array_equal(row[5:8], arr(n.Pos)) and np.array_equal(row[8:11], arr(n.PoseLie)) and np.array_equal(row[11:14], arr(n.Vel)), (i, j)
            assert row[27] == 1.0                  # Kp = 1 with stereo (rebvo_second_t.cpp:486)
            assert int(row[-1]) == int(nms[j][i]), (i, j, row[-1], nms[j])

    # (b) against the reference's own stereo frame order, object by object
    for i in (0, 3):
        orc = oracle.Oracle("ref", oracle.euroc_params(W, H))
        orc.enable_stereo(pc["ppx"], pc["ppy"], pc["zfx"], pc["zfy"], T_PAIR, R_PAIR, 100.0)
        path, prev = 0.0, None
        for k in range(n_fr):
            _, nav = orc.process_frame_stereo(frames[tri(k + i, pool)], pairs[tri(k + i, pool)], t0 + dt * k)
            if k == 0:
                prev = nav
                continue
            row = dumps[i][k - 1]
            kl = orc.keylines((k - 1) % 8)
            assert int(row[2]) == len(kl)
            if k - 1 > 0:
                path += np.linalg.norm(prev.V[:])
                assert np.allclose(row[5:8], prev.Pos[:], atol=1e-6 * path + 1e-9)
                assert abs(int(row[-1]) - prev.pad0) <= max(2, prev.pad0 // 500), (i, k, row[-1], prev.pad0)
            assert abs(row[14] - kl["rho"].sum()) <= 1e-6 * abs(kl["rho"].sum()) + 1e-9
            prev = nav
        orc.close()


def test_stereo_data_sets_as_one_batch_group(tmp_path):
    """DataSetCam members with a stereo pair in one batch group (multi_device_replay --group): the feeder thread of each member puts
    the cam0 list's images into the member's camera ring and the cam1 list's into its pair ring (REBVO::initPairCamera,
    rebvo_first_t.cpp:64-76, 183-199); two data sets of different lengths (a member leaves when its lists end).  Every sequence's dump
    must equal, number for number, the dump of the same data set replayed by an object of its own (its own thread and one-sequence
    context with the rig: the path test_host_stereo_replay checks against the reference)."""
    PIL = pytest.importorskip("PIL.Image")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "rebvo_amd", "lib", "multi_device_replay")
    if not os.path.exists(exe):
        pytest.fail("needs multi_device_replay — a broken snapshot, not a reason to skip: run __graft_entry__.build()")
    p, frames, pairs, pc = make_data(all_pairs=True, nf=8)
    cfgs, lens = [], (8, 6)
    for i, n in enumerate(lens):
        t_ns = [1403636579763555584 + 7 * i + 50_000_000 * k for k in range(n)]
        dirs = {}
        for cam, imgs in (("cam0", frames[i:i + n] if i + n <= 8 else frames[:n]), ("cam1", pairs[i:i + n] if i + n <= 8 else pairs[:n])):
            d = tmp_path / f"seq{i}" / "mav0" / cam
            (d / "data").mkdir(parents=True)
            with open(d / "data.csv", "w") as f:
                f.write("#timestamp [ns],filename\n")
                for k, fr in enumerate(imgs):
                    PIL.fromarray(fr[:, :, 0], "L").save(d / "data" / f"{t_ns[k]}.png")
                    f.write(f"{t_ns[k]},{t_ns[k]}.png\n")
            dirs[cam] = d
        cfg = tmp_path / f"cfg{i}"
        write_global_config(cfg, edgehip.euroc_params(W, H), camera_type=2,
                            dataset=(str(dirs["cam0"] / "data") + "/", str(dirs["cam0"] / "data.csv"), 1e-9),
                            stereo=dict(dir=str(dirs["cam1"] / "data") + "/", file=str(dirs["cam1"] / "data.csv"), **pc))
        cfgs.append(str(cfg))
    for tag, extra in (("alone", []), ("group", ["--group"])):
        r = subprocess.run([exe, "--devices", "1", "--dump", str(tmp_path / tag), *extra, *cfgs], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
        for i, n in enumerate(lens):
            assert f"sequence {i} device 0: frames delivered {n - 1}" in r.stdout, r.stdout[-2000:]
    for i, n in enumerate(lens):
        a = np.loadtxt(str(tmp_path / "alone") + f"{i}.txt", ndmin=2)
        g = np.loadtxt(str(tmp_path / "group") + f"{i}.txt", ndmin=2)
        assert a.shape == g.shape == (n - 1, 14) and np.array_equal(a, g), (i, a, g)
        assert np.abs(a[2:, 5:8]).max() > 0          # poses that moved


def test_stereo_matches_log_needs_a_stereo_context_a_log_and_logged_frames():
    """edgehip_read_stereo_matches_log fails loudly where it has nothing to return: a context without stereo_available, a stereo context
    without a log, a frame that was never enqueued — and works across a ring wrap (frames older than the ring are refused, as for the nav log)."""
    p, frames, pairs, pc = make_data(all_pairs=True, nf=6)
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=1, nslots=3)
    eh.set_nav_log(4)
    with pytest.raises(edgehip.EdgeHipError, match="stereo_available"):
        eh.read_stereo_matches_log(0, 1)
    eh.close()
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H, stereo_available=1), nseq=2, nslots=4)
    eh.set_slot_camera(3, pc["ppx"], pc["ppy"], pc["zfx"], pc["zfy"])
    eh.set_stereo_rig(3, T_PAIR, R_PAIR, 100.0)
    with pytest.raises(edgehip.EdgeHipError):
        eh.read_stereo_matches_log(0, 1)                      # no log yet
    eh.set_nav_log(4)
    with pytest.raises(edgehip.EdgeHipError, match="no frame"):
        eh.read_stereo_matches_log(0, 1)
    seen = []
    for k in range(6):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[k]] * 2))
        eh.upload_rgb(3, np.stack([pairs[k]] * 2))
        eh.process_frame(0.05 * k)
        seen.append(eh.get_stereo_matches().copy())
    got = eh.read_stereo_matches_log(2, 4)                     # frames 2..5: the ring's whole length
    assert got.shape == (4, 2)
    for j, k in enumerate(range(2, 6)):
        assert np.array_equal(got[j], seen[k]), (k, got[j], seen[k])
    assert int(got[-1][0]) > 500 and got[-1][0] == got[-1][1]
    with pytest.raises(edgehip.EdgeHipError, match="not in the log"):
        eh.read_stereo_matches_log(1, 1)                      # overwritten by frame 5
    with pytest.raises(edgehip.EdgeHipError):
        eh.read_stereo_matches_log(6, 1)                      # never enqueued
    eh.close()
