#!/usr/bin/env python3
"""64 sequences through the C-ABI, every frame crossing PCIe (edgehip_upload_rgb_pinned + edgehip_process_frame back to back, no host
waits): what the device's own copy / compute pipeline delivers, without the plugin surface's threads.  One measurement, not a test."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rebvo_amd import edgehip, synth

W, H, N = 752, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = [f for f, _, _ in synth.billboard_sequence(W, H, 12, seed=11)]


def tri(k, n):
    p = 2 * (n - 1); r = k % p
    return r if r < n else p - r


for nslots in (3,):
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=N, nslots=nslots, device=0)
    bufs = [eh.alloc_pinned_frames() for _ in range(4)]
    for b, (arr, _) in enumerate(bufs):
        for s in range(N):
            arr[s] = frames[tri(b + s, 12)]
    eh.set_nav_log(8)
    base = [0]
    for mode in ("upload every frame", "upload every frame, record of frame k-2 read behind frame k", "upload every frame, wait for its copy, record of frame k-2 read behind frame k",
                 "look-ahead like the group: process k, upload k+1, wait copy k, record k-2", "no upload (the slot's old frame again)"):
        def step(k):
            if mode.startswith("look-ahead"):
                if k == 0: eh.upload_rgb_pinned(eh.next_slot(), bufs[tri(k, 4)][1])
                slot = eh.next_slot()
                eh.process_frame(0.05 * k)
                eh.upload_rgb_pinned(eh.next_slot(), bufs[tri(k + 1, 4)][1])
                eh.lib.edgehip_upload_wait(eh.ctx, slot)
                if k - base[0] >= 2: eh.read_nav_log_array(k - 2, 1)
                return
            if mode.startswith("upload"):
                eh.upload_rgb_pinned(eh.next_slot(), bufs[tri(k, 4)][1])
                if "wait" in mode: eh.lib.edgehip_upload_sync(eh.ctx)
            eh.process_frame(0.05 * k)
            if "record" in mode and k - base[0] >= 2: eh.read_nav_log_array(k - 2, 1)
        eh.reset() if hasattr(eh, "reset") else None
        eh.set_nav_log(8)
        for k in range(12): step(k)
        eh.sync()
        t0 = time.perf_counter()
        for k in range(12, 72): step(k)
        eh.sync()
        dt = (time.perf_counter() - t0) / 60
        print(f"nslots {nslots}  {mode:50s} {dt * 1e3:.3f} ms per step, {N / dt:.0f} frames/s", flush=True)
    eh.close()
