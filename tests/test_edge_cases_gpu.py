"""GPU edge cases the reference handles in-band (failure detection): empty edge maps, a scene cut
(too few matches -> estimation restart, rebvo_second_t.cpp:412-422), a batch whose sequences are in different
states at the same time, and the largest supported width."""
import numpy as np
import pytest

from rebvo_amd import edgehip, synth

pytestmark = pytest.mark.gpu


def _oracle(w, h, **kw):
    from oracle import oracle
    kind = "ref" if oracle.available("ref") else "port"
    return oracle.Oracle(kind, oracle.euroc_params(w, h, **kw))


def _check(ng, nr, k):
    assert (ng.kn, ng.tresh) == (nr.kn, nr.tresh), k
    if k == 0:
        return
    assert ng.estimation_ok == nr.estimation_ok, k
    assert abs(ng.klm_num - nr.klm_num) <= max(2, nr.klm_num // 1000), k
    Vr, Wr = np.array(nr.V[:]), np.array(nr.W[:])
    if np.all(np.isfinite(Vr)):
        step = np.linalg.norm(Vr) + np.linalg.norm(Wr)
        assert np.allclose(ng.V[:], Vr, rtol=0, atol=1e-6 * step + 1e-9), (k, ng.V[:], Vr)
        assert np.allclose(ng.W[:], Wr, rtol=0, atol=1e-6 * step + 1e-9), (k, ng.W[:], Wr)
    else:
        assert np.array_equal(np.isnan(ng.V[:]), np.isnan(Vr))


def test_empty_frames_then_texture():
    """Uniform frames give kn == 0 (Minimizer_RV returns at once, EstimateReScalingOpt returns 1, nothing to match);
    then texture appears.  Every frame must agree with the reference, including the restart flags."""
    w, h = 376, 240
    tex = [f for f, _, _ in synth.billboard_sequence(w, h, 4)]
    flat = np.full((h, w, 3), 90, np.uint8)
    frames = [flat, flat, tex[0], tex[1], flat, tex[2], tex[3]]
    orc = _oracle(w, h)
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
    for k, f in enumerate(frames):
        _, nr = orc.process_frame(f, 0.05 * k)
        eh.upload_rgb(eh.next_slot(), f)
        eh.process_frame(0.05 * k)
        _check(eh.read_nav()[0], nr, k)
    _, mask = eh.download_keylines(0, eh.cur_slot())
    assert np.array_equal(mask, orc.mask(orc.cur_slot()))
    eh.close()


@pytest.mark.parametrize("gmt", [500, 20000], ids=["default_threshold", "always_restart"])
def test_scene_cut_and_restart(gmt):
    """A hard scene cut, once with the shipped GlobalMatchThreshold and once with a threshold no frame can reach, so
    that every frame takes the restart branch (V = 0, P_V = 1e50*I, Kp = 1, P_Kp = 10, EstimationOK = false)."""
    w, h = 376, 240
    a = [f for f, _, _ in synth.billboard_sequence(w, h, 4, seed=3)]
    b = list(synth.rects_sequence(w, h, 4))
    frames = a + b                                     # cut between frame 3 and 4
    orc = _oracle(w, h, global_match_threshold=gmt)
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h, global_match_threshold=gmt), nseq=1, nslots=3)
    oks = []
    for k, f in enumerate(frames):
        _, nr = orc.process_frame(f, 0.05 * k)
        eh.upload_rgb(eh.next_slot(), f)
        eh.process_frame(0.05 * k)
        ng = eh.read_nav()[0]
        _check(ng, nr, k)
        if k:   # the first frame has no estimate on either side
            assert abs(ng.Kp - nr.Kp) < 1e-8 and (ng.RKp == nr.RKp or abs(ng.RKp - nr.RKp) <= 1e-8 * abs(nr.RKp))
        oks.append(nr.estimation_ok)
    if gmt > 10000:
        assert not any(oks[1:])
    eh.close()


def test_batch_with_sequences_in_different_states():
    """Three sequences in one launch: textured, empty, and a scene cut — each must equal its own reference run."""
    w, h, n = 376, 240, 6
    tex = [f for f, _, _ in synth.billboard_sequence(w, h, n)]
    flat = np.full((h, w, 3), 40, np.uint8)
    rects = list(synth.rects_sequence(w, h, n))
    seqs = [tex, [flat] * n, tex[:3] + rects[3:]]
    orcs = [_oracle(w, h) for _ in seqs]
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=3, nslots=3)
    for k in range(n):
        eh.upload_rgb(eh.next_slot(), np.stack([s[k] for s in seqs]))
        eh.process_frame(0.05 * k)
        navs = eh.read_nav()
        for s in range(3):
            _, nr = orcs[s].process_frame(seqs[s][k], 0.05 * k)
            _check(navs[s], nr, (s, k))
    eh.close()


def test_max_width_1024_and_max_points_cap():
    """w = 1024 is the widest frame the context accepts; MaxPoints small enough to truncate (kl_max, edge_finder.cpp:203-209)."""
    w, h = 1024, 64
    frames = list(synth.rects_sequence(w, h, 3))
    orc = _oracle(w, h, max_points=600, reference_points=500)
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h, max_points=600, reference_points=500), nseq=1, nslots=3)
    for k, f in enumerate(frames):
        _, nr = orc.process_frame(f, 0.05 * k)
        eh.upload_rgb(eh.next_slot(), f)
        eh.process_frame(0.05 * k)
        ng = eh.read_nav()[0]
        assert (ng.kn, ng.tresh) == (nr.kn, nr.tresh), k
    assert orc.kn(orc.cur_slot()) == 600, "case must hit the MaxPoints cap"
    _, mask = eh.download_keylines(0, eh.cur_slot())
    assert np.array_equal(mask, orc.mask(orc.cur_slot()))
    eh.close()


def test_pinned_upload_equals_staged_upload():
    """edgehip_upload_rgb_pinned (page-locked source, no staging copy) puts the same frames on the device as
    edgehip_upload_rgb: stage A results are identical, for a sub-range of the batch too."""
    w, h = 376, 240
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 3)]
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=3, nslots=2)
    arr, ptr = eh.alloc_pinned_frames()
    arr[0], arr[1], arr[2] = frames[0], frames[1], frames[2]
    eh.upload_rgb(0, np.stack(frames))
    eh.upload_rgb_pinned(1, ptr)
    eh.stage_a(0)
    st0 = eh.get_state(0)
    kns = eh.get_kn(0).copy()
    ref = [eh.download_keylines(s, 0)[0] for s in range(3)]
    # same detector state for the second run
    for s in range(3):
        st = eh.get_state(s)
        st.tresh, st.l_kl_num = edgehip.euroc_params(w, h).detector_thresh, 0
        eh.set_state(s, st)
    eh.stage_a(1)
    assert np.array_equal(eh.get_kn(1), kns) and kns.min() > 3000
    for s in range(3):
        k1 = eh.download_keylines(s, 1)[0]
        for f in ("p_inx", "m_m", "c_p", "n_m"):
            assert np.array_equal(k1[f], ref[s][f]), (s, f)
    eh.sync()
    eh.free_pinned(ptr)
    eh.close()


def test_bound_pool_frames_equal_copied_frames():
    """edgehip_bind_rgb_indexed (stage A reads each sequence's frame in place from a device pool) against
    edgehip_upload_rgb_indexed (the same selection copied into the slot) over a few frames of the whole path, including
    the last frame of the pool (whose pixels are read as 8-byte words into the 16 B of slack) and a switch back to
    ordinary uploads."""
    import torch
    w, h, npool = 376, 240, 5
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, npool)]
    host = np.stack(frames)
    pool = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
    pool[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
    pool[host.size:] = 255
    a = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=3, nslots=3)
    b = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=3, nslots=3)
    for k in range(6):
        idx = np.array([(k + s) % npool for s in (0, 2, 4)], np.int32)
        if k < 5:
            a.bind_rgb_indexed(a.next_slot(), pool.data_ptr(), npool, idx)
        else:
            a.upload_rgb(a.next_slot(), np.stack([frames[i] for i in idx]))   # back to the slot's own storage
        b.upload_rgb_indexed(b.next_slot(), pool.data_ptr(), npool, idx)
        a.process_frame(0.05 * k)
        b.process_frame(0.05 * k)
        for na, nb in zip(a.read_nav(), b.read_nav()):
            assert na.kn == nb.kn and na.tresh == nb.tresh
            assert na.V[:] == nb.V[:] and na.W[:] == nb.W[:] and na.Pos[:] == nb.Pos[:], k
    for s in range(3):
        ka, ma = a.download_keylines(s, a.cur_slot())
        kb, mb = b.download_keylines(s, b.cur_slot())
        assert np.array_equal(ma, mb) and np.array_equal(ka["rho"], kb["rho"]) and len(ka) > 3000
    a.close(); b.close()


@pytest.mark.parametrize("source", ["bound_pool", "pinned_upload"])
def test_replay_without_host_sync_runs_ahead_of_the_device(source):
    """A replay that never synchronises between frames (bench.py's loop: bind the frames, process, next) lets the host
    get more than 8 frames ahead of the device, past the depth of the pinned time-stamp / frame-index rings.  Every
    frame must still see its own time stamp and its own pool indices: the nav log of two sequences equals the CPU
    reference run on the same frame order (the reference is fed synchronously)."""
    import torch
    w, h, npool, nf, B = 752, 480, 6, 14, 4
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, npool, seed=11)]
    host = np.stack(frames)
    pool = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
    pool[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
    torch.cuda.synchronize()
    tri = lambda k: (k % (2 * (npool - 1))) if (k % (2 * (npool - 1))) < npool else 2 * (npool - 1) - (k % (2 * (npool - 1)))
    t = lambda k: 0.05 * k + 0.003 * (k % 3)   # uneven stamps: a stale ring entry would change dt
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=B, nslots=3)
    eh.set_nav_log(nf)
    pinned = []
    for k in range(nf):
        idx = np.array([tri(k + s) for s in range(B)], np.int32)
        if source == "bound_pool":
            eh.bind_rgb_indexed(eh.next_slot(), pool.data_ptr(), npool, idx)
        else:
            # page-locked frames go over the upload stream, under stage A and B/C of the frames before: the copy into a
            # slot must wait for the frame that used the slot three frames earlier, and stage A for the copy
            arr, ptr = eh.alloc_pinned_frames()
            arr[...] = np.stack([frames[i] for i in idx])
            pinned.append(ptr)
            eh.upload_rgb_pinned(eh.next_slot(), ptr)
        eh.process_frame(t(k))
    log = eh.read_nav_log(0, nf)
    for ptr in pinned:
        eh.free_pinned(ptr)
    for s in (0, B - 1):
        orc = _oracle(w, h)
        for k in range(nf):
            _, nr = orc.process_frame(frames[tri(k + s)], t(k))
            ng = log[k][s]
            assert ng.kn == nr.kn and ng.frame == k, (s, k, ng.kn, nr.kn)
            assert np.allclose(ng.V[:], nr.V[:], rtol=0, atol=1e-9) and np.allclose(ng.W[:], nr.W[:], rtol=0, atol=1e-9), (s, k)
            assert np.allclose(ng.Pos[:], nr.Pos[:], rtol=0, atol=1e-9), (s, k)
            assert k == 0 or abs(ng.dt - nr.dt) < 1e-12, (s, k, ng.dt, nr.dt)   # frame 0 has no predecessor
    eh.close()


def test_create_that_runs_out_of_memory_leaves_nothing_behind():
    """edgehip_create for a batch no GPU can hold fails with an error code half-way through its allocations; the partial
    context is torn down (device memory returns to what it was) and the next create works."""
    import torch
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    with pytest.raises(edgehip.EdgeHipError):
        edgehip.EdgeHip(edgehip.euroc_params(752, 480), nseq=400000, nslots=3)
    free1, _ = torch.cuda.mem_get_info()
    assert abs(free0 - free1) < (64 << 20), (free0, free1)
    eh = edgehip.EdgeHip(edgehip.euroc_params(376, 240), nseq=1, nslots=2)
    eh.close()


def test_failed_create_does_not_poison_the_next_frames():
    """Regression (round 1, GPUTEST_r01): HIP keeps a thread's last error until it is read, and the launch checks read
    that slot — the hipMalloc that failed inside an out-of-memory edgehip_create was reported, as "out of memory", by
    the first kernel launch of the NEXT context in the process.  Failed create -> successful create -> frames, in one
    process and one thread, checked against the reference."""
    w, h = 256, 192
    with pytest.raises(edgehip.EdgeHipError):
        edgehip.EdgeHip(edgehip.euroc_params(752, 480), nseq=400000, nslots=3)
    orc = _oracle(w, h)
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
    for k, (f, _, _) in enumerate(synth.billboard_sequence(w, h, 3)):
        _, nr = orc.process_frame(f, 0.05 * k)
        eh.upload_rgb(eh.next_slot(), f)
        eh.process_frame(0.05 * k)            # round 1: "hipGetLastError() -> out of memory" here
        _check(eh.read_nav()[0], nr, k)
    # a foreign error left in the slot by someone else's HIP call must not surface either
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 60)) != 0      # fails, stays in the last-error slot
    f = next(iter(synth.billboard_sequence(w, h, 1)))[0]
    eh.upload_rgb(eh.next_slot(), f)
    eh.process_frame(0.2)
    eh.stage_a(eh.cur_slot())
    eh.close()


def test_two_slots_bound_between_frames_keep_their_own_indices():
    """Regression: edgehip_bind_rgb_indexed staged the index vector of every bind between two process_frame
    calls in the same pinned row, so a second bind (the stereo pair slot, a prefetch of the next frame) could overwrite
    the indices of the first while its asynchronous copy was still pending."""
    import torch
    w, h, npool, B = 256, 192, 6, 64
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, npool, seed=5)]
    host = np.stack(frames)
    pool = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
    pool[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
    torch.cuda.synchronize()
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h, auto_gain=0.0), nseq=B, nslots=3)   # fixed threshold: kn depends on the frame only
    ia = np.array([s % npool for s in range(B)], np.int32)
    ib = np.array([(s + 3) % npool for s in range(B)], np.int32)
    for rep in range(20):                     # the race needs the first copy to be still pending: try a few times
        eh.bind_rgb_indexed(0, pool.data_ptr(), npool, ia)
        eh.bind_rgb_indexed(1, pool.data_ptr(), npool, ib)
        eh.stage_a(0)
        eh.stage_a(1)
        kn0, kn1 = eh.get_kn(0), eh.get_kn(1)
        if rep == 0:
            # what each pool frame gives on its own
            ref = []
            for i in range(npool):
                eh.bind_rgb_indexed(2, pool.data_ptr(), npool, np.full(B, i, np.int32))
                eh.stage_a(2)
                ref.append(int(eh.get_kn(2)[0]))
            assert len(set(ref)) > 1, ref     # the frames are distinguishable by their KeyLine count
        assert [int(v) for v in kn0] == [ref[i] for i in ia], rep
        assert [int(v) for v in kn1] == [ref[i] for i in ib], rep
    eh.close()


def test_rebinding_a_slot_without_a_sync_does_not_disturb_the_stage_a_in_flight():
    """Regression: stage A reads a binding's index row in place from page-locked memory.  bind(slot, ia);
    stage_a(slot); bind(slot, ib) with no synchronisation in between writes the SAME row (frames_seen has not moved) while the
    first stage A may still be reading it — the second bind has to wait for it.  Large batch, so that the kernel is still
    running when the second bind arrives; the KeyLine counts of the first stage A must be those of `ia`."""
    import torch
    w, h, npool, B = 256, 192, 6, 256
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, npool, seed=5)]
    host = np.stack(frames)
    pool = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
    pool[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
    torch.cuda.synchronize()
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h, auto_gain=0.0), nseq=B, nslots=3)
    ref = []
    for i in range(npool):
        eh.bind_rgb_indexed(2, pool.data_ptr(), npool, np.full(B, i, np.int32))
        eh.stage_a(2)
        ref.append(int(eh.get_kn(2)[0]))
    assert len(set(ref)) > 1, ref
    ia = np.array([s % npool for s in range(B)], np.int32)
    ib = np.array([(s + 3) % npool for s in range(B)], np.int32)
    for rep in range(10):
        eh.bind_rgb_indexed(0, pool.data_ptr(), npool, ia)
        eh.stage_a(0)
        eh.bind_rgb_indexed(0, pool.data_ptr(), npool, ib)      # no sync since the stage A above
        # the slot's KeyLine counts are still those of the first stage A (nothing has run on the new binding yet)
        assert [int(v) for v in eh.get_kn(0)] == [ref[i] for i in ia], rep
        eh.stage_a(0)
        assert [int(v) for v in eh.get_kn(0)] == [ref[i] for i in ib], rep
    eh.close()


def test_nav_log_reads_outside_the_logged_frames_are_refused():
    """Regression: edgehip_read_nav_log returned stale or zero records with rc 0 for frames that were never
    enqueued, or that a frame one ring length later had already overwritten."""
    w, h = 188, 120
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 6)]
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=2, nslots=3)
    eh.set_nav_log(4)
    with pytest.raises(RuntimeError):
        eh.read_nav_log_array(0, 1)                    # nothing enqueued yet
    for k in range(3):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[k]] * 2))
        eh.process_frame(0.05 * k)
    assert list(eh.read_nav_log_array(0, 3)["frame"][:, 0]) == [0, 1, 2]
    with pytest.raises(RuntimeError, match="not in the log"):
        eh.read_nav_log_array(1, 3)                    # frame 3 has not been enqueued
    for k in range(3, 6):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[k]] * 2))
        eh.process_frame(0.05 * k)
    assert list(eh.read_nav_log_array(2, 4)["frame"][:, 1]) == [2, 3, 4, 5]
    with pytest.raises(RuntimeError, match="not in the log"):
        eh.read_nav_log_array(1, 2)                    # frame 1 left the 4-entry ring when frame 5 was written
    eh.close()


@pytest.mark.parametrize("kind", ["noise", "shuffled"])
def test_frames_the_tracker_cannot_explain_finish_in_bounded_time(kind):
    """White noise, and a scene that jumps by a hundred pixels every frame: the minimiser diverges, velocity estimates go
    wild, and every data-dependent loop of the path (search_match's walk above all) must still terminate quickly —
    a frame of four small sequences takes ~1 ms, the bound is 1 s."""
    import time
    from rebvo_amd import synth
    w, h, n = 376, 240, 4
    rs = np.random.RandomState(11)
    scene = [f for f, _, _ in synth.billboard_sequence(w, h, 4)]
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=n, nslots=3)
    try:
        for k in range(8):
            if kind == "noise":
                fr = rs.randint(0, 256, (n, h, w, 3), dtype=np.uint8)
            else:
                fr = np.stack([np.roll(scene[(k + s) % 4], 97 * k * (s + 1), axis=1) for s in range(n)])
            eh.upload_rgb(eh.next_slot(), np.ascontiguousarray(fr))
            t0 = time.perf_counter()
            eh.process_frame(0.05 * k)
            eh.sync()
            assert time.perf_counter() - t0 < 1.0, (kind, k)
        assert all(x.kn > 0 for x in eh.read_nav())
    finally:
        eh.close()


def test_upload_wait_hands_one_slots_buffer_back_while_the_next_copy_runs():
    """edgehip_upload_wait(slot): the page-locked source of the copies into THAT slot has been read — the copy into the next slot,
    enqueued behind it, may still run.  The batch group's order (rebvo_amd/host/src/batch_group.cpp): process frame k, send frame
    k+1 up, wait for frame k's copy, give its buffer back — here the buffer is scribbled over at once, and the records must equal
    the plain upload_rgb / process_frame run bit for bit."""
    w, h, B, nf = 376, 240, 6, 9
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 5, seed=23)]
    tri = lambda k: (k % 8) if (k % 8) < 5 else 8 - (k % 8)
    batch = lambda k: np.stack([frames[tri(k + s)] for s in range(B)])
    ref = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=B, nslots=3)
    ref.set_nav_log(nf)
    for k in range(nf):
        ref.upload_rgb(ref.next_slot(), batch(k))
        ref.process_frame(0.05 * k)
    want = ref.read_nav_log_array(0, nf)
    ref.close()
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=B, nslots=3)
    eh.set_nav_log(nf)
    bufs = [eh.alloc_pinned_frames() for _ in range(2)]
    bufs[0][0][...] = batch(0)
    eh.upload_rgb_pinned(eh.next_slot(), bufs[0][1])
    for k in range(nf):
        slot = eh.next_slot()
        eh.process_frame(0.05 * k)
        if k + 1 < nf:
            bufs[(k + 1) % 2][0][...] = batch(k + 1)
            eh.upload_rgb_pinned(eh.next_slot(), bufs[(k + 1) % 2][1])
        assert eh.lib.edgehip_upload_wait(eh.ctx, slot) == 0
        bufs[k % 2][0][...] = 0x5a            # the application writes its next frame here
    got = eh.read_nav_log_array(0, nf)
    assert got.tobytes() == want.tobytes()
    assert eh.lib.edgehip_upload_wait(eh.ctx, 7) != 0   # no such slot
    for _, ptr in bufs:
        eh.free_pinned(ptr)
    eh.close()
