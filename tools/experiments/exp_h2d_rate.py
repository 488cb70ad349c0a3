#!/usr/bin/env python3
"""Host-to-device rate of edgehip_upload_rgb_pinned for a group-sized copy (64 RGB24 frames = 69 MB), alone and while CPU threads
write into the page-locked buffers the way the application's copyFrom does.  tools/experiments: one measurement, not a test."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rebvo_amd import edgehip

W, H, N = 752, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 64
eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=N, nslots=3, device=0)
bufs = [eh.alloc_pinned_frames() for _ in range(4)]
src = np.random.default_rng(0).integers(0, 255, (12, H, W, 3), dtype=np.uint8)
for arr, _ in bufs:
    for s in range(N):
        arr[s] = src[s % 12]
nbytes = N * H * W * 3


def rate(label, reps=40):
    eh.lib.edgehip_upload_sync(eh.ctx)
    t0 = time.perf_counter()
    for k in range(reps):
        eh.upload_rgb_pinned(k % 3, bufs[k % 4][1])
        eh.lib.edgehip_upload_sync(eh.ctx)
    dt = time.perf_counter() - t0
    print(f"{label:58s} {nbytes * reps / dt / 1e9:6.1f} GB/s  ({dt / reps * 1e3:.3f} ms per {nbytes / 1e6:.0f} MB copy)", flush=True)


rate("warm-up")
rate("alone")
stop = False


def writer(tid, T, same):
    k = 0
    while not stop:
        arr = bufs[(k + (0 if same else 2)) % 4][0]
        for s in range(tid, N, T):
            arr[s] = src[(s + k) % 12]
        k += 1


for T, same in ((8, False), (8, True), (32, False)):
    stop = False
    th = [threading.Thread(target=writer, args=(t, T, same)) for t in range(T)]
    for t in th: t.start()
    time.sleep(0.2)
    rate(f"with {T} CPU threads writing {'the same' if same else 'other'} page-locked buffers")
    stop = True
    for t in th: t.join()
rate("alone again")
