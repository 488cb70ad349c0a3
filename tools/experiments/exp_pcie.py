"""PCIe-inclusive rate: every step uploads fresh host frames through edgehip_upload_rgb (memcpy into the pinned staging
buffer + one H2D copy on the stage-A stream) and runs the whole path.  usage: exp_pcie.py [nseq]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from rebvo_amd import edgehip, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = [f for f, _, _ in synth.billboard_sequence(752, 480, 12)]
batches = [np.ascontiguousarray(np.stack([f] * B)) for f in frames]
eh = edgehip.EdgeHip(edgehip.euroc_params(), nseq=B, nslots=3)
pinned = [eh.alloc_pinned_frames() for _ in range(2)]
for (arr, ptr), b in zip(pinned, batches):
    arr[...] = b
for mode in ("h2d only", "h2d + path", "pinned only", "pinned+path", "pin+memcpy+p"):
    for rep in range(2):
        eh.reset()
        eh.sync()
        t0 = time.time()
        for k, b in enumerate(batches):
            slot = eh.next_slot() if "path" in mode or mode.endswith("+p") else k % 3
            if mode.startswith("h2d"):
                eh.upload_rgb(slot, b)
            else:
                arr, ptr = pinned[k % 2]
                if mode == "pin+memcpy+p":      # the application writes the frames into the pinned buffer itself (camera DMA
                    if k >= 2:                  # or decoder output would land there directly); double-buffered
                        eh.sync()
                    arr[...] = b
                eh.upload_rgb_pinned(slot, ptr)
            if "path" in mode or mode.endswith("+p"):
                eh.process_frame(0.05 * k)
        eh.sync()
        dt = (time.time() - t0) / len(batches)
    gb = B * 752 * 480 * 3 / 1e9
    print(f"{mode:12s} B={B}: {dt*1e3:7.2f} ms/step  {B/dt:9.0f} frames/s  {gb/dt:6.1f} GB/s host->device")
