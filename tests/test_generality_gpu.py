"""GPU: the image sizes and detector windows the reference's containers take and the HIP path used to refuse
(VERDICT r3 "missing" 3 and 4): widths that are not a multiple of 4, widths above 1024 columns, DetectorPlaneFitSize 1 and 3.
iimage / sspace / edge_finder are sized from Size2D and build_mask builds PInv for any win_s (iimage.cpp:53-128, sspace.cpp:52-60,
edge_finder.cpp:69-100, 110-137); all of it bit-exact against oracle/_ref, like tests/test_stage_a_gpu.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from rebvo_amd import edgehip, synth
from tests.test_stage_a_gpu import STAGE_A_FIELDS, _bits, _run

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("w,h", [(754, 480), (1280, 720), (2048, 64), (1026, 50), (757, 131), (17, 23)])
def test_stage_a_any_width(w, h):
    """w % 4 in {0, 1, 2}: rows of the RGB frame and of every plane start off a 16-byte boundary; above 1024 columns the
    detector's LDS planes hold 8 (1280) or 4 (2048) rows per band instead of 12."""
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 2)] if w >= 200 and h >= 100 else list(synth.rects_sequence(w, h, 2, seed=4))
    kn = _run(w, h, frames)
    if w * h > 300_000:
        assert kn > 5000


@pytest.mark.parametrize("w,h,nseq", [(754, 480, 1), (1280, 720, 2)])
def test_pipeline_at_other_widths(w, h, nseq):
    """The whole path (field tiles, tracker gathers, matching walks, mapper) on an image whose width is not a multiple of 4 /
    is above 1024 columns: per-frame tolerance of tests/test_pipeline_gpu.py."""
    from tests.test_pipeline_gpu import _run as run_pipeline
    run_pipeline(w, h, 5, nseq=nseq, min_kn=5000)


def test_grey8_and_pool_ingest_at_an_odd_pixel_count():
    """757 x 131 = 99 167 pixels: frames of a mono pool and of an RGB pool start off every alignment; the 8-bit expansion and the
    pool gather take their byte-wise forms.  Same KeyLines as the plain RGB24 upload."""
    import torch
    w, h = 757, 131
    frames = list(synth.rects_sequence(w, h, 3, seed=6))
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=2, nslots=3)
    eh.upload_rgb(0, np.stack([frames[1], frames[2]]))
    eh.stage_a(0)
    want = [eh.download_keylines(s, 0) for s in range(2)]
    assert len(want[0][0]) > 50 and want[0][0].tobytes() != want[1][0].tobytes()
    mono = np.stack([np.ascontiguousarray(f[:, :, 0]) for f in frames])
    rgb = np.stack([np.repeat(m[:, :, None], 3, 2) for m in mono])
    eh.upload_rgb(0, np.stack([rgb[1], rgb[2]]))
    eh.stage_a(0)
    want = [eh.download_keylines(s, 0) for s in range(2)]
    pools = {}
    for name, host in (("grey8", mono), ("rgb", rgb)):
        t = torch.empty(host.size + 16, dtype=torch.uint8, device="cuda")
        t[:host.size] = torch.from_numpy(host.reshape(-1)).cuda()
        pools[name] = t
    torch.cuda.synchronize()
    idx = np.array([1, 2], np.int32)
    for feed in ("upload_grey8", "bind_grey8", "bind_rgb", "upload_rgb_indexed"):
        if feed == "upload_grey8":
            eh.upload_grey8(1, np.stack([mono[1], mono[2]]))
        elif feed == "bind_grey8":
            eh.bind_grey8_indexed(1, pools["grey8"].data_ptr(), 3, idx)
        elif feed == "bind_rgb":
            eh.bind_rgb_indexed(1, pools["rgb"].data_ptr(), 3, idx)
        else:
            eh.upload_rgb_indexed(1, pools["rgb"].data_ptr(), 3, idx)
        eh.stage_a(1)
        for s in range(2):
            kl, mask = eh.download_keylines(s, 1)
            assert kl.tobytes() == want[s][0].tobytes() and np.array_equal(mask, want[s][1]), (feed, s)
    eh.close()


@pytest.mark.parametrize("ws", [1, 3])
@pytest.mark.parametrize("w,h", [(376, 240), (754, 480)])
def test_stage_a_plane_fit_windows(tmp_path, ws, w, h):
    """DetectorPlaneFitSize 1 (3x3 window) and 3 (7x7): the sign balance over the window, the plane fit with the 3 x n pseudo
    inverse built as the reference builds it, the scanned area [ws, dim - ws).  The reference runs in a process of its own
    (it caches PInv for the first window size it sees)."""
    from oracle import oracle
    if not oracle.available("ref"):
        pytest.fail("oracle/_ref not built" " — a broken snapshot, not a reason to skip: run __graft_entry__.build()")
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 3)]
    over = {"plane_fit_size": ws}
    inp, outp = tmp_path / "in.npz", tmp_path / "out.npz"
    np.savez(inp, frames=np.stack(frames), w=w, h=h, over=json.dumps(over))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stage_a_ref_runner.py"), str(inp), str(outp)],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ref = np.load(outp)
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h, debug_planes=1, **over), nseq=1, nslots=3)
    kns = []
    for k, f in enumerate(frames):
        slot = k % 3
        eh.upload_rgb(slot, f)
        eh.stage_a(slot)
        kl, mask = eh.download_keylines(0, slot)
        st = eh.get_state(0)
        assert len(kl) == int(ref[f"kn_{k}"]), (k, len(kl), int(ref[f"kn_{k}"]))
        assert st.tresh == float(ref[f"tresh_{k}"]) and st.l_kl_num == int(ref[f"lkl_{k}"])
        assert np.array_equal(_bits(eh.download_plane(0, "dog")), _bits(ref[f"dog_{k}"]))
        assert np.array_equal(mask, ref[f"mask_{k}"]), f"frame {k}: img_mask_kl differs"
        rk = ref[f"kl_{k}"]
        for fld in STAGE_A_FIELDS:
            assert np.array_equal(kl[fld], rk[fld]), f"frame {k}: KeyLine.{fld} differs"
        assert np.float32(st.retuned_thresh) == np.float32(ref[f"retuned_{k}"])
        kns.append(len(kl))
    eh.close()
    assert min(kns) > 1000
    # the window does change the result: not the 5x5 KeyLines under another name
    eh2 = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=1, nslots=3)
    eh2.upload_rgb(0, frames[0])
    eh2.stage_a(0)
    assert not np.array_equal(eh2.download_keylines(0, 0)[1], ref["mask_0"])
    eh2.close()


def test_create_limits():
    with pytest.raises(RuntimeError, match="width"):
        edgehip.EdgeHip(edgehip.euroc_params(2052, 64), nseq=1, nslots=2)
    with pytest.raises(RuntimeError, match="DetectorPlaneFitSize"):
        edgehip.EdgeHip(edgehip.euroc_params(376, 240, plane_fit_size=4), nseq=1, nslots=2)
