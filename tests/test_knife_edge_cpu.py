"""The frame on which bench.py's heterogeneous sequence 5 leaves the reference (BENCH_r02: 1.45 % of the path), pinned on
the CPU: the reference (oracle/_ref) and our restatement (oracle/port) agree to 1e-15 for eight frames and differ by
8.2e-7 on the ninth.  Not the 6x6 solve (round 2's guess): the systems of the init phase have condition numbers of 36..49,
nowhere near TooN::SVD's 1e9 cut-off, and the reference with another dgesvd_ behind TooN stays where it was.  It is ONE
KeyLine, detected exactly on a half pixel (c_p.x = 124.5), whose re-projection at X = 0 — the first evaluation of every
Minimizer_RV with TrackerInitType 2 — rounds to pixel 124 or 125 depending on the last bits of its depth
(oracle.half_pixel_keylines explains the mechanism); the field holds a KeyLine at one of the two pixels and nothing at
the other."""
import numpy as np
import pytest

from helpers import hetero_sequence
from oracle import oracle

pytestmark = pytest.mark.skipif(not (oracle.available("ref") and oracle.available("port")), reason="needs both CPU oracles")

W, H, SEQ, SPLIT = 752, 480, 5, 8


def _run(kind, backend=None, nframes=SPLIT + 1):
    o = oracle.Oracle(kind, oracle.euroc_params(W, H))
    if backend is not None:
        prev = o.svd_backend(backend)
    frame_of = hetero_sequence(SEQ)
    navs, last_trace, amb = [], None, None
    for k in range(nframes):
        old = o.keylines(o.cur_slot()).copy() if k else None
        o.svd_trace_start()
        _, n = o.process_frame(frame_of(k), 0.05 * k)
        last_trace = o.svd_trace_stop()
        navs.append((np.array(n.V[:]), np.array(n.W[:]), int(n.klm_num)))
        if k == nframes - 1:
            amb = oracle.half_pixel_keylines(old, o.field(o.cur_slot())[:, :, 1], o.p.ppx, o.p.ppy, n.s_rho_q, W, H)
    if backend is not None:
        o.svd_backend(prev)
    o.close()
    return navs, last_trace, amb


def test_the_split_frame_is_a_half_pixel_keyline_not_an_ill_conditioned_solve():
    ref, tr_ref, amb = _run("ref")
    port, tr_port, amb_p = _run("port")
    for k in range(1, SPLIT):                        # eight frames in agreement
        assert np.max(np.abs(ref[k][0] - port[k][0])) < 1e-13 and np.max(np.abs(ref[k][1] - port[k][1])) < 1e-13, k
    d = max(np.max(np.abs(ref[SPLIT][0] - port[SPLIT][0])), np.max(np.abs(ref[SPLIT][1] - port[SPLIT][1])))
    assert 1e-8 < d < 1e-5, d                        # the ninth: 8.2e-7
    # the init phase's four systems (two per trial), reference side: well conditioned, nothing near the cut-off
    assert len(tr_ref) == 4 and len(tr_port) == 4
    for r in tr_ref:
        s = np.sort(np.abs(r["s"]))[::-1]
        assert s[0] / s[5] < 1e3
    # the very first system (zero-initialised trial, evaluated at X = 0 from bit-equal KeyLine geometry) already differs by
    # far more than rounding: a match decision, not arithmetic
    a0, b0 = tr_ref[0]["A"], tr_port[0]["A"]
    assert np.max(np.abs(a0 - b0)) > 1e-6 * np.max(np.abs(a0))
    # and the detector names the KeyLine: exactly on a half pixel, two different field entries at its candidate pixels
    assert [i for i, _ in amb] == [13558] and amb == amb_p
    assert amb[0][1] == [-1, 13177]


def test_the_reference_stays_put_when_dgesvd_changes():
    """Same reference code with the harness's one-sided Jacobi behind dgesvd_ instead of MKL: 1e-15, also on the split frame."""
    a, _, _ = _run("ref", backend=0)
    b, _, _ = _run("ref", backend=1)
    for k in range(1, SPLIT + 1):
        assert np.max(np.abs(a[k][0] - b[k][0])) < 1e-13 and np.max(np.abs(a[k][1] - b[k][1])) < 1e-13, k
        assert a[k][2] == b[k][2]
