"""8-bit mono ingest (GPU): edgehip_upload_grey8 / _pinned / edgehip_bind_grey8_indexed against the RGB24 path fed the
r = g = b expansion of the same image — what the reference's DataSetCam hands ConvertRGB2BW for a mono data set
(datasetcam.cpp:109-171, image.h:197-203).  b + g + r = 3 v either way, so everything downstream must be bit-identical:
kn, img_mask_kl, every KeyLine field, the poses of a short replay.  Both stage-A paths: the one-kernel path reads the 8-bit
frame in its first load (SRC_GREY8), the multi-kernel path gets the RGB24 expansion on the device."""
import os

import numpy as np
import pytest
import torch

from helpers import require_ref
from rebvo_amd import edgehip, synth

pytestmark = pytest.mark.gpu


def _mono_frames(w, h, n, seed=4):
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, n, seed=seed)]
    mono = [np.ascontiguousarray(f[:, :, 1]) for f in frames]
    rgb = [np.ascontiguousarray(np.repeat(m[:, :, None], 3, axis=2)) for m in mono]
    return mono, rgb


def _replay(w, h, feed, nf, nseq=2):
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=nseq, nslots=3)
    navs = []
    for k in range(nf):
        feed(eh, eh.next_slot(), k)
        eh.process_frame(0.05 * k)
        navs.append(eh.read_nav())
    kl, mask = eh.download_keylines(nseq - 1, eh.cur_slot())
    eh.close()
    return navs, kl, mask


def _same(a, b):
    na, kla, ma = a
    nb, klb, mb = b
    assert np.array_equal(ma, mb)
    assert kla.tobytes() == klb.tobytes()
    for ra, rb in zip(na, nb):
        for x, y in zip(ra, rb):
            assert (x.kn, x.klm_num, x.estimation_ok) == (y.kn, y.klm_num, y.estimation_ok)
            assert bytes(x) == bytes(y)      # the whole record, bit for bit


@pytest.mark.parametrize("mode,w,h", [("3", 752, 480), ("3", 320, 240), ("3", 200, 152), ("1", 376, 240), ("0", 200, 152)],
                         ids=["fused_752", "fused_320", "fused_generic", "multi_kernel", "auto_small_batch"])
def test_grey8_upload_is_bit_identical_to_the_rgb24_expansion(mode, w, h, monkeypatch):
    monkeypatch.setenv("EDGEHIP_LEVEL_MODE", mode)
    mono, rgb = _mono_frames(w, h, 4)
    a = _replay(w, h, lambda eh, s, k: eh.upload_rgb(s, np.stack([rgb[k]] * 2)), 4)
    b = _replay(w, h, lambda eh, s, k: eh.upload_grey8(s, np.stack([mono[k]] * 2)), 4)
    _same(a, b)
    assert a[0][-1][0].kn > 500


def test_grey8_against_the_reference_fed_rgb24():
    """The device on 8-bit frames against the CPU reference on their RGB24 expansion: mask and pose."""
    oracle = require_ref()
    w, h = 376, 240
    mono, rgb = _mono_frames(w, h, 4, seed=9)
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h))
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
    for k in range(4):
        _, nr = orc.process_frame(rgb[k], 0.05 * k)
        eh.upload_grey8(eh.next_slot(), mono[k])
        eh.process_frame(0.05 * k)
        ng = eh.read_nav()[0]
        assert (ng.kn, ng.klm_num) == (nr.kn, nr.klm_num)
        if k:
            assert np.allclose(ng.V[:], nr.V[:], rtol=0, atol=1e-9) and np.allclose(ng.W[:], nr.W[:], rtol=0, atol=1e-9)
    _, mask = eh.download_keylines(0, eh.cur_slot())
    assert np.array_equal(mask, orc.mask(orc.cur_slot()))
    eh.close()
    orc.close()


def test_grey8_pinned_and_bound_pool_variants(monkeypatch):
    """Page-locked upload (upload stream) and frames read in place from a device pool, in the one-kernel stage A; and a slot that
    changes format from frame to frame."""
    monkeypatch.setenv("EDGEHIP_LEVEL_MODE", "3")
    w, h, nf = 752, 480, 4
    mono, rgb = _mono_frames(w, h, nf, seed=6)
    ref = _replay(w, h, lambda eh, s, k: eh.upload_rgb(s, np.stack([rgb[k]] * 2)), nf)

    pinned = {}

    def feed_pinned(eh, s, k):
        if "buf" not in pinned:
            pinned["buf"] = [eh.alloc_pinned_grey8() for _ in range(2)]
        arr, ptr = pinned["buf"][k % 2]
        eh.sync()
        arr[...] = np.stack([mono[k]] * 2)
        eh.upload_grey8_pinned(s, ptr)
    _same(ref, _replay(w, h, feed_pinned, nf))

    host = np.stack(mono)
    pool = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
    pool[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
    _same(ref, _replay(w, h, lambda eh, s, k: eh.bind_grey8_indexed(s, pool.data_ptr(), nf, np.array([k, k], np.int32)), nf))

    def feed_mixed(eh, s, k):
        if k % 2:
            eh.upload_grey8(s, np.stack([mono[k]] * 2))
        else:
            eh.upload_rgb(s, np.stack([rgb[k]] * 2))
    _same(ref, _replay(w, h, feed_mixed, nf))


@pytest.mark.parametrize("mode", ["3", "1"], ids=["fused", "multi_kernel"])
def test_grey8_with_the_undistorting_load(mode, monkeypatch):
    """UseUndistort (BASELINE configs[3]): the 8-bit frame is expanded on the device and resampled like the RGB24 one."""
    monkeypatch.setenv("EDGEHIP_LEVEL_MODE", mode)
    w, h = 640, 480
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 3, fx=525.0, fy=525.0, cx=320.0, cy=240.0, seed=3)]
    mono = [np.ascontiguousarray(f[:, :, 0]) for f in frames]
    p = edgehip.tum_params(w, h, use_undistort=1)
    res = []
    for feed in ("rgb", "grey"):
        eh = edgehip.EdgeHip(p, nseq=1, nslots=3)
        navs = []
        for k in range(3):
            if feed == "rgb":
                eh.upload_rgb(eh.next_slot(), frames[k])
            else:
                eh.upload_grey8(eh.next_slot(), mono[k])
            eh.process_frame(0.02 * k)
            navs.append(bytes(eh.read_nav()[0]))
        kl, mask = eh.download_keylines(0, eh.cur_slot())
        res.append((navs, kl.tobytes(), mask))
        eh.close()
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2])


def test_undistorted_download_of_a_mono_slot():
    """Regression: edgehip_download_undistorted took every slot for RGB24 — for a slot fed 8-bit mono frames
    (uploaded, or bound to a pool of mono frames) it resampled the wrong bytes and, for a bound pool, read past its end.  The
    mono slot's download must equal the RGB24 slot's (r = g = b = v), whether or not stage A has run on it."""
    import torch
    w, h = 640, 480
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 3, fx=525.0, fy=525.0, cx=320.0, cy=240.0, seed=3)]
    mono = [np.ascontiguousarray(f[:, :, 0]) for f in frames]
    p = edgehip.tum_params(w, h, use_undistort=1)
    eh = edgehip.EdgeHip(p, nseq=2, nslots=3)
    eh.upload_rgb(0, np.stack([frames[0], frames[1]]))
    want = [eh.download_undistorted(s, 0) for s in range(2)]
    assert not np.array_equal(want[0], want[1])
    eh.upload_grey8(1, np.stack([mono[0], mono[1]]))
    for s in range(2):
        assert np.array_equal(eh.download_undistorted(s, 1), want[s])     # straight after the upload, before any stage A
    eh.stage_a(1)
    for s in range(2):
        assert np.array_equal(eh.download_undistorted(s, 1), want[s])
    host = np.stack(mono)
    pool = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
    pool[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
    torch.cuda.synchronize()
    eh.bind_grey8_indexed(2, pool.data_ptr(), 3, np.array([1, 0], np.int32))     # index 2 * n * 3 would lie outside the mono pool
    assert np.array_equal(eh.download_undistorted(0, 2), want[1]) and np.array_equal(eh.download_undistorted(1, 2), want[0])
    eh.close()
