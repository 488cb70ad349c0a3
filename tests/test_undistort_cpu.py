"""CPU-only: the undistortion map libedgehip builds on the host (edgehip_build_undistort_map, no device
needed) against the reference's own image_undistort ctor (src/VideoLib/image_undistort.cpp:29-95) run
through oracle/_ref.  Integer taps and 16.16 weights must be identical, including pixels whose taps fall
outside the image (pincushion coefficients push the corners out)."""
import numpy as np
import pytest

from oracle import oracle
from rebvo_amd import edgehip

CASES = {
    "tum_euroc_kc": dict(),                                             # barrel, all taps valid
    "pincushion": dict(kc=(0.35, 0.1, 0.0, 1e-3, -2e-3)),               # corners sample outside: num < 4, num == 0
    "euroc_native": dict(w=752, h=480),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_undistort_map_matches_reference(name):
    if not oracle.available("ref"):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    over = dict(CASES[name])
    w, h = over.pop("w", 640), over.pop("h", 480)
    kc = over.pop("kc", None)
    mk_o = oracle.euroc_params if w == 752 else oracle.tum_params
    mk_e = edgehip.euroc_params if w == 752 else edgehip.tum_params
    po, pe = mk_o(w, h, use_undistort=1), mk_e(w, h, use_undistort=1)
    if kc is not None:
        po.kc[:] = kc
        pe.kc[:] = kc
    orc = oracle.Oracle("ref", po)
    inx_r, iw_r = orc.undistort_map()
    inx_g, iw_g = edgehip.build_undistort_map(pe)
    if name == "pincushion":
        assert (inx_r < 0).any(), "case must exercise invalid taps"
    assert np.array_equal(inx_r, inx_g)
    assert np.array_equal(iw_r, iw_g)
