"""Shared helpers of the GPU parity tests."""
import numpy as np


def to_edgehip_kl(kl):
    """oracle KEYLINE_DTYPE and edgehip KEYLINE_DTYPE are the same 168-byte layout."""
    from rebvo_amd import edgehip
    return np.frombuffer(np.ascontiguousarray(kl).tobytes(), dtype=edgehip.KEYLINE_DTYPE).copy()


def oracle_pair(w, h, n_warm, seq="billboard", **over):
    """Run the reference oracle for n_warm full frames, then stage A of the next frame.

    Returns (orc, slot_old, slot_new, nav_of_last_full_frame, frames)."""
    from oracle import oracle
    from rebvo_amd import synth
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h, **over))
    if seq == "billboard":
        frames = [f for f, _, _ in synth.billboard_sequence(w, h, n_warm + 1)]
    else:
        frames = list(synth.rects_sequence(w, h, n_warm + 1))
    nav = None
    for k in range(n_warm):
        _, nav = orc.process_frame(frames[k], 0.05 * k)
    slot_old = (n_warm - 1) % 8
    slot_new = n_warm % 8
    orc.stage_a(slot_new, frames[n_warm], nav.tresh, nav.kn)
    return orc, slot_old, slot_new, nav, frames


def inject_pair(eh, orc, slot_old, slot_new, seq=0, gslot_old=0, gslot_new=1):
    eh.upload_keylines(seq, gslot_old, to_edgehip_kl(orc.keylines(slot_old)), orc.mask(slot_old), orc.retuned(slot_old))
    eh.upload_keylines(seq, gslot_new, to_edgehip_kl(orc.keylines(slot_new)), orc.mask(slot_new), orc.retuned(slot_new))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
