"""GPU: the fused stage-A kernel (stage_a_fused.hip, EDGEHIP_LEVEL_MODE=3; the default from 32 sequences per launch on at widths 752 / 640, 64 at 320, 192 at others; EDGEHIP_FUSED_MIN_BATCH
overrides) against the reference, with the bar of the multi-kernel path it replaces: img0 / img1 / DoG / gradient
planes, img_mask_kl, kn, every KeyLine field stage A defines, the threshold state and reTunedThresh — all bit-exact.
Sizes cover: the bench size, heights that are not a multiple of the 4-row tick, widths that are not a multiple of 64
or 16, an image narrower than one column group, the kl_max truncation, empty images, and a batch of different frames."""
import os

import numpy as np
import pytest

from rebvo_amd import edgehip, synth
import test_stage_a_gpu as tsa
import test_level_kernel_gpu as tlk

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def fused_mode():
    old = os.environ.get("EDGEHIP_LEVEL_MODE")
    os.environ["EDGEHIP_LEVEL_MODE"] = "3"
    yield
    if old is None:
        del os.environ["EDGEHIP_LEVEL_MODE"]
    else:
        os.environ["EDGEHIP_LEVEL_MODE"] = old


SIZES = [(192, 144), (200, 150), (752, 480), (376, 240), (640, 480), (100, 36), (768, 64), (896, 64), (64, 19), (16, 16)]


@pytest.mark.parametrize("w,h", SIZES, ids=[f"{w}x{h}" for w, h in SIZES])
def test_fused_planes_mask_keylines_bit_exact(w, h):
    if w >= 300:
        frames = [f for f, _, _ in synth.billboard_sequence(w, h, 3)]
    else:
        frames = list(synth.rects_sequence(w, h, 3, seed=3))
    tsa._run(w, h, frames)


@pytest.mark.parametrize("w,h", [(752, 480), (640, 480), (320, 240)], ids=["euroc_752", "tum_640", "default_config_320"])
def test_fused_product_instantiations_without_debug_planes(w, h):
    """The shipped widths have their own instantiation (compile-time LDS offsets, no debug-plane code): mask, kn, KeyLines,
    threshold state against the reference."""
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 4, seed=2)]
    tsa._run(w, h, frames, check_planes=False)


def test_fused_kl_max_truncation():
    frames = list(synth.rects_sequence(320, 240, 2, seed=9))
    tsa._run(320, 240, frames, over=dict(max_points=700, reference_points=600, track_points=600))


def test_fused_empty_image():
    f = np.full((144, 192, 3), 77, np.uint8)
    tsa._run(192, 144, [f, f])


def test_fused_noise_image_many_candidates():
    """White noise: nearly every pixel passes the gradient gate, so the per-wave candidate lists run at their capacity."""
    rng = np.random.default_rng(5)
    frames = [np.repeat(rng.integers(0, 256, (120, 256, 1), dtype=np.uint8), 3, axis=2) for _ in range(2)]
    tsa._run(256, 120, frames, over=dict(max_points=50000, reference_points=30000, track_points=12000, min_thresh=1e-4,
                                         detector_thresh=1e-4))


def test_fused_batch_of_different_frames():
    tlk._scale_space_case("euroc_752x480", 752, 480, False, {})


def test_fused_is_the_default_for_large_batches_and_matches_the_multi_kernel_path():
    """160 sequences (>= EDGEHIP_FUSED_MIN_BATCH) in auto mode against the same batch on the forced multi-kernel path."""
    w, h, B = 376, 240, 160
    pool = [f for f, _, _ in synth.billboard_sequence(w, h, 5)]
    outs = []
    for mode in ("0", "2"):
        os.environ["EDGEHIP_LEVEL_MODE"] = mode
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=B, nslots=3)
        for k in range(3):
            eh.upload_rgb(eh.next_slot(), np.stack([pool[(k + s) % 5] for s in range(B)]))
            eh.process_frame(0.05 * k)
        navs = eh.read_nav()
        kls = [eh.download_keylines(s, eh.cur_slot()) for s in (0, 1, B - 1)]
        outs.append((navs, kls))
        eh.close()
    (na, ka), (nb, kb) = outs
    for x, y in zip(na, nb):
        assert x.kn == y.kn and x.tresh == y.tresh and x.V[:] == y.V[:] and x.W[:] == y.W[:] and x.Pos[:] == y.Pos[:]
    for (k1, m1), (k2, m2) in zip(ka, kb):
        assert np.array_equal(m1, m2) and k1.tobytes() == k2.tobytes()


def test_fused_with_the_undistorting_source():
    """UseUndistort (BASELINE config 4): image_undistort + ConvertRGB2BW in k_undistort_grey, then the fused kernel on the 16-bit
    grey plane — planes, mask and KeyLines against the reference, with the reference's own undistortion map semantics."""
    tlk._scale_space_case("tum_undistort_640x480", 640, 480, True, {})


def test_fused_with_the_undistortion_inside_its_load(monkeypatch):
    """EDGEHIP_FUSED_UNDIST=1 (BASELINE config 4 as north_star words it): the four taps and their 16.16 weights gathered by the
    one-kernel stage A itself (SRC_UNDIST), no k_undistort_grey and no 16-bit plane — planes, mask and KeyLines against the
    reference.  Measured slower than the pre-pass (DESIGN.md section 3a''), so it is an option, not the default."""
    from tests.helpers import needs_experiments
    needs_experiments()
    monkeypatch.setenv("EDGEHIP_FUSED_UNDIST", "1")
    tlk._scale_space_case("tum_undistort_640x480", 640, 480, True, {})
    tlk._scale_space_case("tum_undistort_320x240", 320, 240, True, {})


@pytest.mark.parametrize("in_load", ["0", "1"])
def test_fused_tum_product_instantiation_matches_the_multi_kernel_path(monkeypatch, in_load):
    """TUM 640x480 + undistort at a batch that takes the fused path by default (the W = 640 / grey-plane instantiation, no debug
    planes; in_load = 1: the W = 640 instantiation that resamples inside its load) against the same batch on the forced multi-kernel
    path: identical nav records and depth maps."""
    if in_load == "1":
        from tests.helpers import needs_experiments
        needs_experiments()
    monkeypatch.setenv("EDGEHIP_FUSED_UNDIST", in_load)
    w, h, B = 640, 480, 192
    pool = [f for f, _, _ in synth.billboard_sequence(w, h, 4, fx=525.0, fy=525.0, cx=320.0, cy=240.0)]
    outs = []
    for mode in ("0", "2"):
        os.environ["EDGEHIP_LEVEL_MODE"] = mode
        eh = edgehip.EdgeHip(edgehip.tum_params(w, h, use_undistort=1), nseq=B, nslots=3)
        for k in range(3):
            eh.upload_rgb(eh.next_slot(), np.stack([pool[(k + s) % 4] for s in range(B)]))
            eh.process_frame(0.05 * k)
        navs = eh.read_nav()
        kls = [eh.download_keylines(s, eh.cur_slot()) for s in (0, 1, B - 1)]
        outs.append((navs, kls))
        eh.close()
    (na, ka), (nb, kb) = outs
    for x, y in zip(na, nb):
        assert x.kn == y.kn and x.kn > 5000 and x.tresh == y.tresh and x.V[:] == y.V[:] and x.W[:] == y.W[:] and x.Pos[:] == y.Pos[:]
    for (k1, m1), (k2, m2) in zip(ka, kb):
        assert np.array_equal(m1, m2) and k1.tobytes() == k2.tobytes()
