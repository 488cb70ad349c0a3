"""Whole-frame parity (GPU): edgehip_process_frame vs the reference's FirstThr+SecondThread sequencing.

No state is injected: both sides start from the same RGB frames and run N frames.  Stage A stays bit-exact
for as long as the auto-threshold state agrees (it does: kn is an integer and the P-controller is exact).
Pose tolerance: |dV|,|dW| <= 1e-6 relative to the step size (+1e-9 abs) per frame — fp32-level, the bound
the north star asks for — and the integrated position within 1e-6 of the path length.
"""
import numpy as np
import pytest

from rebvo_amd import edgehip, synth
from helpers import depths_agree

pytestmark = pytest.mark.gpu


def _run(w, h, n, nseq=1, over=None, min_kn=0):
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    over = over or {}
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n)]
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h, **over))
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h, **over), nseq=nseq, nslots=3)
    path = 0.0
    for k, f in enumerate(frames):
        _, nr = orc.process_frame(f, 0.05 * k)
        eh.upload_rgb(eh.next_slot(), np.stack([f] * nseq))
        eh.process_frame(0.05 * k)
        for ng in eh.read_nav():
            assert ng.kn == nr.kn, f"frame {k}: kn {ng.kn} vs {nr.kn}"
            assert ng.tresh == nr.tresh
            if k == 0:
                continue
            assert ng.estimation_ok == nr.estimation_ok
            Vr, Wr = np.array(nr.V[:]), np.array(nr.W[:])
            step = np.linalg.norm(Vr) + np.linalg.norm(Wr)
            assert np.allclose(ng.V[:], Vr, rtol=0, atol=1e-6 * step + 1e-9), (k, ng.V[:], Vr)
            assert np.allclose(ng.W[:], Wr, rtol=0, atol=1e-6 * step + 1e-9), (k, ng.W[:], Wr)
            assert abs(ng.klm_num - nr.klm_num) <= max(2, nr.klm_num // 1000), (k, ng.klm_num, nr.klm_num)
            assert abs(ng.s_rho_q - nr.s_rho_q) <= 1e-9
            assert abs(ng.Kp - nr.Kp) < 1e-8
            path += np.linalg.norm(Vr)
            assert np.allclose(ng.Pos[:], nr.Pos[:], atol=1e-6 * path + 1e-9)
            assert np.allclose(ng.Pose[:], nr.Pose[:], atol=1e-7)
    # final depth maps agree
    slot = eh.cur_slot()
    kg, mask = eh.download_keylines(0, slot)
    kr = orc.keylines(orc.cur_slot())
    assert np.array_equal(mask, orc.mask(orc.cur_slot()))
    same = kg["m_id"] == kr["m_id"]
    assert same.mean() > 0.999
    assert depths_agree(kg, kr, same)
    assert np.allclose(kg["s_rho"][same], kr["s_rho"][same], rtol=1e-5, atol=1e-7)
    assert len(kg) >= min_kn, len(kg)
    eh.close()


def test_pipeline_small():
    _run(376, 240, 8)


def test_pipeline_many_keylines():
    """More than 16384 KeyLines in one sequence: the block tables of the minimiser's steps exceed what their one-go prefetch
    covers (more than 64 blocks), and k_rescale's small-batch form streams what neither its registers nor its LDS hold."""
    _run(752, 480, 9, over=dict(max_points=30000, reference_points=27000, track_points=24000, detector_thresh=0.005,
                                min_thresh=1e-4), min_kn=17000)


def test_pipeline_euroc_size():
    _run(752, 480, 6)


def test_pipeline_batched_sequences_identical():
    _run(376, 240, 4, nseq=3)


def test_framecount_follows_reference_ring():
    """MatchNumThresh > 0 makes TryVelRot skip KeyLines with m_num < min(MatchNumThresh, FrameCount)
    (global_tracker.cpp:356); FrameCount lives in the reference's 8 PipeBuffer slots, so the value a frame
    sees depends on THAT ring length, whatever nslots the GPU context uses."""
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    w, h, n = 376, 240, 19
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n)]
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h, match_num_thresh=2))
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h, match_num_thresh=2), nseq=1, nslots=3)
    for k, f in enumerate(frames):
        _, nr = orc.process_frame(f, 0.05 * k)
        eh.upload_rgb(eh.next_slot(), f)
        eh.process_frame(0.05 * k)
        ng = eh.read_nav()[0]
        assert ng.kn == nr.kn
        if k:
            step = np.linalg.norm(nr.V[:]) + np.linalg.norm(nr.W[:])
            assert np.allclose(ng.V[:], nr.V[:], rtol=0, atol=1e-6 * step + 1e-9), k
            assert np.allclose(ng.W[:], nr.W[:], rtol=0, atol=1e-6 * step + 1e-9), k
    eh.close()


def test_depth_reset_semantics():
    """edgehip_depth_reset == REBVO::Reset() as SecondThread runs it (rebvo_second_t.cpp:609-620)."""
    w, h = 376, 240
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 4)]
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=2, nslots=3)
    for k, f in enumerate(frames):
        eh.upload_rgb(eh.next_slot(), np.stack([f, f]))
        eh.process_frame(0.05 * k)
    before = eh.get_state(0)
    eh.depth_reset(1)
    s0, s1 = eh.get_state(0), eh.get_state(1)
    assert list(s0.Pos[:]) == list(before.Pos[:]) and np.linalg.norm(s0.Pos[:]) > 0   # sequence 0 untouched
    assert list(s1.Pos[:]) == [0, 0, 0] and list(s1.V[:]) == [0, 0, 0] and list(s1.W[:]) == [0, 0, 0]
    assert np.array_equal(np.array(s1.Pose[:]).reshape(3, 3), np.eye(3))
    assert s1.tresh == s0.tresh                                 # detector state carries on
    k0, _ = eh.download_keylines(0, eh.cur_slot())
    k1, _ = eh.download_keylines(1, eh.cur_slot())
    assert (k1["rho"] == 1.0).all() and (k1["s_rho"] == 20.0).all()
    assert (k0["s_rho"] < 20.0).any()
    eh.close()


def test_pipeline_with_stream_overlap(monkeypatch):
    """EDGEHIP_OVERLAP=1: stage A of frame k+1 runs on its own stream under stages B/C of frame k (the reference's
    T0 || T1 pipelining).  Ordering is by events only, so the results must not change: same parity bars, and the
    records must equal the serialised run bit for bit."""
    monkeypatch.setenv("EDGEHIP_OVERLAP", "1")
    _run(376, 240, 8)
    _run(376, 240, 5, nseq=3)
    frames = [f for f, _, _ in synth.billboard_sequence(376, 240, 6)]

    def records(overlap):
        monkeypatch.setenv("EDGEHIP_OVERLAP", overlap)
        eh = edgehip.EdgeHip(edgehip.euroc_params(376, 240), nseq=2, nslots=3)
        out = []
        for k, f in enumerate(frames):
            eh.upload_rgb(eh.next_slot(), np.stack([f, frames[(k + 1) % 6]]))
            eh.process_frame(0.05 * k)            # no read-back in between: frames are enqueued back to back
        for n in eh.read_nav():
            out.append((n.kn, n.klm_num, tuple(n.V[:]), tuple(n.W[:]), tuple(n.Pos[:]), n.tresh))
        kl, mask = eh.download_keylines(1, eh.cur_slot())
        eh.close()
        return out, kl.tobytes(), mask.tobytes()

    assert records("0") == records("1")


def test_reset_mid_sequence_matches_reference():
    """REBVO::Reset() (depth reset of the newest edge map + pose/velocity reset, rebvo_second_t.cpp:609-620) after frame 3:
    the following frames must track like the reference does after the same reset."""
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    w, h, n = 376, 240, 8
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n)]
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
    for k, f in enumerate(frames):
        _, nr = orc.process_frame(f, 0.05 * k)
        eh.upload_rgb(eh.next_slot(), f)
        eh.process_frame(0.05 * k)
        ng = eh.read_nav()[0]
        assert ng.kn == nr.kn
        if k:
            step = np.linalg.norm(nr.V[:]) + np.linalg.norm(nr.W[:])
            assert np.allclose(ng.V[:], nr.V[:], rtol=0, atol=1e-6 * step + 1e-9), k
            assert np.allclose(ng.W[:], nr.W[:], rtol=0, atol=1e-6 * step + 1e-9), k
            assert np.allclose(ng.Pos[:], nr.Pos[:], atol=1e-7), k
            assert abs(ng.klm_num - nr.klm_num) <= max(2, nr.klm_num // 1000), k
        if k == 3:
            orc.depth_reset()
            eh.depth_reset()
    kg, _ = eh.download_keylines(0, eh.cur_slot())
    kr = orc.keylines(orc.cur_slot())
    same = kg["m_id"] == kr["m_id"]
    assert same.mean() > 0.999 and depths_agree(kg, kr, same)
    eh.close()


def test_pipeline_batch_of_different_sequences():
    """Three DIFFERENT sequences in one batch (different scenes and trajectories, different KeyLine counts, one of them
    starting from a blank frame): every sequence must follow its own reference run — nothing leaks between the
    per-sequence slices of the batched buffers."""
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build(), where the reference tree is present")
    w, h, n = 376, 240, 7
    seqs = [[f for f, _, _ in synth.billboard_sequence(w, h, n, seed=11, traj_seed=13)],
            [f for f, _, _ in synth.billboard_sequence(w, h, n, seed=5, traj_seed=3)],
            [np.repeat(f[:, :, None], 3, axis=2) if f.ndim == 2 else f for f in synth.rects_sequence(w, h, n)]]
    seqs[2][0] = np.full_like(seqs[2][0], 40)          # a blank first frame: no KeyLines, then a hard start
    orcs = [oracle.Oracle("ref", oracle.euroc_params(w, h)) for _ in seqs]
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=3, nslots=3)
    kns = set()
    for k in range(n):
        refs = [orc.process_frame(s[k], 0.05 * k)[1] for orc, s in zip(orcs, seqs)]
        eh.upload_rgb(eh.next_slot(), np.stack([s[k] for s in seqs]))
        eh.process_frame(0.05 * k)
        for ng, nr in zip(eh.read_nav(), refs):
            assert ng.kn == nr.kn and ng.tresh == nr.tresh, k
            kns.add(nr.kn)
            if k == 0:
                continue
            assert ng.estimation_ok == nr.estimation_ok
            Vr, Wr = np.array(nr.V[:]), np.array(nr.W[:])
            step = np.linalg.norm(Vr) + np.linalg.norm(Wr)
            assert np.allclose(ng.V[:], Vr, rtol=0, atol=1e-6 * step + 1e-9), (k, ng.V[:], Vr)
            assert np.allclose(ng.W[:], Wr, rtol=0, atol=1e-6 * step + 1e-9)
            assert abs(ng.klm_num - nr.klm_num) <= max(2, nr.klm_num // 1000)
    assert len(kns) > 2 * n          # the sequences really differ
    for s, orc in enumerate(orcs):
        kg, mask = eh.download_keylines(s, eh.cur_slot())
        kr = orc.keylines(orc.cur_slot())
        assert np.array_equal(mask, orc.mask(orc.cur_slot()))
        same = kg["m_id"] == kr["m_id"]
        assert same.mean() > 0.995
        assert depths_agree(kg, kr, same)
    eh.close()


def test_pipeline_with_frame_graphs(monkeypatch):
    """EDGEHIP_GRAPH=1: the frame's launches are captured into HIP graphs (24 variants: ring slot x FrameCount row) and
    replayed; 30 frames so that every variant is captured AND replayed, results as without graphs."""
    monkeypatch.setenv("EDGEHIP_GRAPH", "1")
    _run(376, 240, 30)


@pytest.mark.parametrize("nseq", [1, 4])
def test_fused_evaluation_and_lm_step_is_bit_identical_to_the_launch_chain(monkeypatch, nseq):
    """Small batches launch every TryVelRot evaluation together with the LM step that follows it (k_try_velrot_lm: the block
    that finishes last runs the step; EDGEHIP_PERSIST_LM = largest batch that does; off by default: it measured no faster).  Same code, same schedule,
    same reduction order as the chain of separate launches: every nav record and the depth map must be identical bit for
    bit — and both follow the reference."""
    from tests.helpers import needs_experiments
    needs_experiments()
    w, h, n = 376, 240, 10
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n + nseq)]
    outs = []
    for mode in ("0", "8"):
        monkeypatch.setenv("EDGEHIP_PERSIST_LM", mode)
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=nseq, nslots=3)
        eh.set_nav_log(n)
        for k in range(n):
            eh.upload_rgb(eh.next_slot(), np.stack([frames[k + s] for s in range(nseq)]))   # a different sequence per slot
            eh.process_frame(0.05 * k)
        log = eh.read_nav_log_array(0, n)
        kl = [eh.download_keylines(s, eh.cur_slot())[0] for s in range(nseq)]
        outs.append((log, kl))
        eh.close()
    (la, ka), (lb, kb) = outs
    assert la.tobytes() == lb.tobytes()
    for x, y in zip(ka, kb):
        assert x.tobytes() == y.tobytes()
    assert np.all(la["estimation_ok"][2:] == 1) and np.all(la["minimizer_evals"][1:] == 12)
    if nseq == 1:
        monkeypatch.setenv("EDGEHIP_PERSIST_LM", "8")
        _run(w, h, 6)     # the fused path against the reference


@pytest.mark.parametrize("nseq,over", [(1, {}), (3, {}), (70, {}), (2, {"tracker_init_iter_num": 0}), (2, {"tracker_init_iter_num": 4})])
def test_two_chain_evaluation_is_bit_identical_to_the_launch_chain(monkeypatch, nseq, over):
    """TrackerInitType = 2: evaluation i of the zero-init chain and of the prior-init chain of Minimizer_RV
    (global_tracker.cpp:649-692 / 698-738) go out as ONE launch (k_try_velrot2 + k_lm_step2, the default) — 9 dependent
    evaluations instead of 12.  Per chain the arithmetic and the summation order are the launch chain's (EDGEHIP_DUAL_INIT=0),
    so every nav record and the whole depth map must agree bit for bit; the evaluation count the records carry stays the
    reference's.  nseq 70: past the batch size where the opening step rides on the preparation launch."""
    w, h, n = 376, 240, 9
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n + min(nseq, 8))]
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("EDGEHIP_DUAL_INIT", mode)
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h, **over), nseq=nseq, nslots=3)
        eh.set_nav_log(n)
        for k in range(n):
            eh.upload_rgb(eh.next_slot(), np.stack([frames[k + s % 8] for s in range(nseq)]))
            eh.process_frame(0.05 * k)
        log = eh.read_nav_log_array(0, n)
        kl = [eh.download_keylines(s, eh.cur_slot())[0] for s in range(min(nseq, 4))]
        outs.append((log, kl))
        eh.close()
    (la, ka), (lb, kb) = outs
    assert la.tobytes() == lb.tobytes()
    for x, y in zip(ka, kb):
        assert x.tobytes() == y.tobytes()
    evals = 2 * (1 + over.get("tracker_init_iter_num", 2)) + 1 + 5
    assert np.all(la["estimation_ok"][2:] == 1) and np.all(la["minimizer_evals"][1:] == evals)


@pytest.mark.parametrize("w,h,nseq,nslots", [(376, 240, 3, 3), (376, 240, 1, 2), (752, 480, 200, 3)])
def test_matching_in_one_pass_is_bit_identical_to_the_three_kernels(monkeypatch, w, h, nseq, nslots):
    """ImuMode 0 without a stereo pair: FordwardMatch's copy, rotate_keylines and directed_matching as k_fwd_win + k_rotate<out of
    place> + k_directed<FUSED> (the default: every new KeyLine's ten matching fields written once, by the kernel that visits it
    anyway) against k_fwd_win + k_fwd_apply + k_rotate + k_directed (EDGEHIP_FUSE_MATCH=0).  The same values travel by another
    route, so every nav record, the new edge map and — after the turned values have been brought into the old slot's own arrays on
    demand — the old edge map must agree bit for bit.  200 sequences: the one-kernel stage A (fill mode) in front; one sequence with
    two slots: the old slot is the one the next frame's detector overwrites; a frame of noise in the middle: a sequence whose tracker
    gives up (no directed matching, the forward copy alone) and restarts."""
    n = 7
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n + min(nseq, 8))]
    rs = np.random.RandomState(5)
    noise = rs.randint(0, 255, size=frames[0].shape).astype(np.uint8)
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("EDGEHIP_FUSE_MATCH", mode)
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=nseq, nslots=nslots)
        eh.set_nav_log(n)
        olds = []
        for k in range(n):
            batch = [frames[k + s % 8] for s in range(nseq)]
            if k == 4:
                batch[0] = noise          # sequence 0 loses track on this frame
            prev = eh.cur_slot()
            eh.upload_rgb(eh.next_slot(), np.stack(batch))
            eh.process_frame(0.05 * k)
            if k in (2, 5):               # the old edge map right after a frame: turned p_m / m_m / rho / s_rho
                olds.append([eh.download_keylines(s, prev)[0] for s in range(min(nseq, 2))])
        log = eh.read_nav_log_array(0, n)
        kl = [eh.download_keylines(s, eh.cur_slot())[0] for s in range(min(nseq, 3))]
        outs.append((log, kl, olds))
        eh.close()
    (la, ka, oa), (lb, kb, ob) = outs
    assert la.tobytes() == lb.tobytes()
    for x, y in zip(ka, kb):
        assert x.tobytes() == y.tobytes()
    for fa, fb in zip(oa, ob):
        for x, y in zip(fa, fb):
            assert len(x) > 1000 and x.tobytes() == y.tobytes()
    assert np.all(la["klm_num"][2:4] > 1000)


def test_stage_level_calls_on_a_slot_the_frame_driver_rotated_out_of_place(monkeypatch):
    """After a whole frame in the one-pass form the old slot's own arrays still hold the unturned KeyLines (the turned values wait
    next to them).  Whatever touches that slot through the stage-level API must see what rotate_keylines would have left in place:
    EstimateQuantile on it, an upload into ONE of its sequences (the other sequences keep their turned KeyLines), a depth reset of
    one sequence, a second rotation.  Each against the same calls on a context running the three-kernel form (EDGEHIP_FUSE_MATCH=0)."""
    w, h, n, nseq = 376, 240, 4, 3
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n + nseq)]
    Rz = np.array([[np.cos(0.01), -np.sin(0.01), 0], [np.sin(0.01), np.cos(0.01), 0], [0, 0, 1.0]])
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("EDGEHIP_FUSE_MATCH", mode)
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=nseq, nslots=3)
        for k in range(n):
            prev = eh.cur_slot()
            eh.upload_rgb(eh.next_slot(), np.stack([frames[k + s] for s in range(nseq)]))
            eh.process_frame(0.05 * k)
        got = {}
        eh.quantile(prev)                                           # s_rho of the OLD slot: turned (scaled by 1 / q_z)
        got["s_rho_q"] = [eh.get_state(s).s_rho_q for s in range(nseq)]
        kl0, m0 = eh.download_keylines(0, prev)
        eh.upload_keylines(1, prev, kl0[:100], m0)                  # sequence 1 of the old slot replaced ...
        got["kl2_after_upload"] = eh.download_keylines(2, prev)[0].tobytes()   # ... sequence 2 keeps its turned KeyLines
        got["kl1_after_upload"] = eh.download_keylines(1, prev)[0].tobytes()
        outs.append(got)
        eh.close()
        # a fresh context for the calls that write the slot before anything has read it
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=nseq, nslots=3)
        for k in range(n):
            prev = eh.cur_slot()
            eh.upload_rgb(eh.next_slot(), np.stack([frames[k + s] for s in range(nseq)]))
            eh.process_frame(0.05 * k)
        eh.depth_reset_slot(prev, 0)
        eh.rotate_keylines(prev, Rz)
        got["after_reset_and_rotation"] = [eh.download_keylines(s, prev)[0].tobytes() for s in range(nseq)]
        eh.close()
    a, b = outs
    assert a["s_rho_q"] == b["s_rho_q"]
    for key in ("kl2_after_upload", "kl1_after_upload", "after_reset_and_rotation"):
        assert a[key] == b[key], key


@pytest.mark.parametrize("w,h,n", [(376, 240, 10), (752, 480, 5)])
def test_reweighted_evaluation_with_two_keylines_per_thread(monkeypatch, w, h, n):
    """The reweighted TryVelRot evaluations with two KeyLines per thread (k_try_velrot_rw2: both KeyLines' gathers in flight
    together, one reduction for the two; EDGEHIP_TVR_RW2 = smallest launch that takes it; measured no faster, so off by default).
    Forced on for a small batch: the path follows the reference inside the usual tolerance, and agrees with the one-KeyLine
    kernel to rounding (the 28 sums are added in another order) with identical discrete results — KeyLine counts, match counts,
    forward matches, EstimationOK."""
    from tests.helpers import needs_experiments
    needs_experiments()
    monkeypatch.setenv("EDGEHIP_TVR_RW2", "1")
    _run(w, h, n, nseq=2)
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n + 2)]
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("EDGEHIP_TVR_RW2", mode)
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=3, nslots=3)
        eh.set_nav_log(n)
        for k in range(n):
            eh.upload_rgb(eh.next_slot(), np.stack([frames[k + s] for s in range(3)]))
            eh.process_frame(0.05 * k)
        outs.append((eh.read_nav_log_array(0, n), [eh.download_keylines(s, eh.cur_slot())[0] for s in range(3)]))
        eh.close()
    (la, ka), (lb, kb) = outs
    for f in ("kn", "klm_num", "klm_fwd", "estimation_ok", "minimizer_evals"):
        assert np.array_equal(la[f], lb[f]), f
    assert np.allclose(la["V"], lb["V"], rtol=1e-9, atol=1e-13) and np.allclose(la["W"], lb["W"], rtol=1e-9, atol=1e-13)
    for x, y in zip(ka, kb):
        assert np.array_equal(x["m_id"], y["m_id"]) and np.allclose(x["rho"], y["rho"], rtol=1e-9, atol=1e-13)
