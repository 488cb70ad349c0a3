"""GPU: the kernels a few sequences take (a live camera: 16 waves per 64 columns in the column prefix, one round trip per
row block in the box average, 1024-thread quantile, the rescaling with 12288 KeyLines in registers, the per-sequence steps
that ride on other launches: frame begin + reEstimateThresh in k_quantile, the minimisation's opening step in
k_tvr_prepare, exp(W) + NaN check in k_fwd_win, the match-count check in k_regularize, pose + nav record in k_rescale)
against the ones whole batches take: the same sequence must come out bit for bit the same whatever the batch it runs in."""
import numpy as np
import pytest

from rebvo_amd import edgehip, synth

pytestmark = pytest.mark.gpu


def _run(w, h, B, pool, nframes, params):
    eh = edgehip.EdgeHip(params, nseq=B, nslots=3)
    navs = []
    for k in range(nframes):
        eh.upload_rgb(eh.next_slot(), np.stack([pool[(k + s) % len(pool)] for s in range(B)]))
        eh.process_frame(0.05 * k)
        navs.append(eh.read_nav())
    kls = [eh.download_keylines(s, eh.cur_slot()) for s in range(min(2, B))]
    eh.close()
    return navs, kls


@pytest.mark.parametrize("w,h,B", [(376, 240, 200), (752, 480, 72), (752, 480, 40), (376, 240, 24), (376, 240, 300)],
                         ids=["one_kernel_stage_a_200", "one_kernel_stage_a_72", "one_kernel_stage_a_40_small_batch_kernels", "multi_kernel_24",
                              "partial_second_round_300"])
def test_a_sequence_does_not_depend_on_the_batch_it_runs_in(w, h, B):
    """2 takes every small-batch kernel.  B = 200 (any width: from 192 sequences on) and 72 (width 752: from 32 on) take the one-kernel stage A (beside the previous frame's tracking below one
    sequence per CU) and every whole-batch kernel (k_rescale / k_quantile / minimiser launches: above 64 sequences); 40 the one-kernel stage A with the
    small-batch forms of those; 24 the multi-kernel stage A; 300 a full round of the one-kernel stage A's workgroups and a partial one, where the library
    again runs stage A beside the previous frame's tracking (api.hip: partial_round)."""
    pool = [f for f, _, _ in synth.billboard_sequence(w, h, 6, seed=4)]
    params = edgehip.euroc_params(w, h)
    n_small, k_small = _run(w, h, 2, pool, 7, params)
    n_big, k_big = _run(w, h, B, pool, 7, params)
    for k, (a, b) in enumerate(zip(n_small, n_big)):
        for s in range(2):
            assert bytes(a[s]) == bytes(b[s]), (k, s, a[s].kn, b[s].kn, a[s].V[:], b[s].V[:], a[s].Kp, b[s].Kp)
    for (ka, ma), (kb, mb) in zip(k_small, k_big):
        assert np.array_equal(ma, mb) and ka.tobytes() == kb.tobytes()
    assert n_small[-1][0].kn > 2000 and n_small[-1][0].estimation_ok == 1


def test_single_sequence_stream_orders_agree(monkeypatch):
    """The three ways a small batch's launches can be ordered: the default (stage A of the next frame on a stream of its own,
    beside this frame's tracking and mapping), EDGEHIP_OVERLAP=0 (one stream, no events) and EDGEHIP_GRAPH=1 (whole-frame graphs
    of the same launches; the page-locked rows of time stamps and bound frame indices are baked into the nodes: one graph per
    ring entry)."""
    w, h = 376, 240
    pool = [f for f, _, _ in synth.billboard_sequence(w, h, 6, seed=4)]
    outs = []
    for env in ({}, {"EDGEHIP_OVERLAP": "0"}, {"EDGEHIP_GRAPH": "1"}):
        for k in ("EDGEHIP_OVERLAP", "EDGEHIP_GRAPH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs.append(_run(w, h, 1, pool, 30, edgehip.euroc_params(w, h)))
    for other in outs[1:]:
        for a, b in zip(outs[0][0], other[0]):
            assert bytes(a[0]) == bytes(b[0])
        assert outs[0][1][0][0].tobytes() == other[1][0][0].tobytes()


def test_whole_batch_with_stage_a_overlap_matches_serial(monkeypatch):
    """EDGEHIP_OVERLAP=1 with the one-kernel stage A: the next frame's detection runs beside this frame's stage B, so nothing of
    stage B may read what that detection overwrites (the modulus histogram and the detector's extremes: reEstimateThresh's tail is
    done by k_quantile only when no such overlap is possible)."""
    w, h, B = 376, 240, 256
    pool = [f for f, _, _ in synth.billboard_sequence(w, h, 6, seed=4)]
    outs = []
    for ov in ("0", "1"):
        monkeypatch.setenv("EDGEHIP_OVERLAP", ov)
        outs.append(_run(w, h, B, pool, 16, edgehip.euroc_params(w, h)))
    for k, (a, b) in enumerate(zip(outs[0][0], outs[1][0])):
        for s_ in range(B):
            assert bytes(a[s_]) == bytes(b[s_]), (k, s_)
