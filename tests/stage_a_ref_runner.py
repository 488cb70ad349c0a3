"""Helper run as its own process: stage A of the reference on a few frames with a DetectorPlaneFitSize other than the 5x5 window.

edge_finder::build_mask caches the plane-fit pseudo inverse in a function-local static sized by the FIRST call's win_s
(edge_finder.cpp:80-100: `static bool PhiReCalc`, `static Matrix<> PInv`), so a process in which the oracle has already run with
another window cannot serve — every window size gets a process of its own.

    stage_a_ref_runner.py <in.npz> <out.npz>     in: frames [n][h][w][3] u8, w, h, over (JSON parameter overrides)
                                                 out: per frame k: kn_k, tresh_k, lkl_k, retuned_k, mask_k, kl_k (KeyLine records)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import oracle
    inp = np.load(sys.argv[1])
    w, h = int(inp["w"]), int(inp["h"])
    over = json.loads(str(inp["over"]))
    orc = oracle.Oracle("ref", oracle.euroc_params(w, h, **over))
    tresh, lkl = orc.p.detector_thresh, 0
    out = {}
    for k, f in enumerate(inp["frames"]):
        slot = k % 8
        kn, tresh, lkl = orc.stage_a(slot, np.ascontiguousarray(f), tresh, lkl)
        out[f"kn_{k}"], out[f"tresh_{k}"], out[f"lkl_{k}"] = kn, tresh, lkl
        out[f"retuned_{k}"] = np.float32(orc.retuned(slot))
        out[f"mask_{k}"] = orc.mask(slot)
        out[f"kl_{k}"] = orc.keylines(slot)
        out[f"dog_{k}"] = orc.plane(slot, "dog")
    np.savez(sys.argv[2], **out)


if __name__ == "__main__":
    main()
