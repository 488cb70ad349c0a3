"""ms per frame of a single sequence (and small batches) with the minimiser as a launch chain / as one persistent launch."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from rebvo_amd import edgehip, synth
w, h = 752, 480
frames = [f for f, _, _ in synth.billboard_sequence(w, h, 12)]
import torch
pool = torch.from_numpy(np.stack(frames).reshape(-1)).cuda()
pool = torch.cat([pool, torch.zeros(16, dtype=torch.uint8, device="cuda")])
tri = lambda k, n: (k % (2 * (n - 1))) if (k % (2 * (n - 1))) < n else 2 * (n - 1) - (k % (2 * (n - 1)))
for nseq in (1, 8):
    for mode in ("0", "8", "0", "8"):
        os.environ["EDGEHIP_PERSIST_LM"] = mode
        eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=nseq, nslots=3)
        def run(k0, n):
            for k in range(k0, k0 + n):
                eh.upload_rgb(eh.next_slot(), np.stack([frames[tri(k, 12)]] * nseq))
                eh.process_frame(0.05 * k)
            eh.sync()
        run(0, 12)
        # frames resident: time process_frame alone
        def run2(k0, n):
            for k in range(k0, k0 + n):
                eh.bind_rgb_indexed(eh.next_slot(), pool.data_ptr(), 12, np.full(nseq, tri(k, 12), np.int32))
                eh.process_frame(0.05 * k)
            eh.sync()
        run2(12, 12)
        t0 = time.perf_counter(); run2(24, 200); dt = time.perf_counter() - t0
        eh.profile_enable(True); run2(224, 20); pr = eh.profile_read()
        print(f"nseq {nseq} EDGEHIP_PERSIST_LM={mode}: {dt / 200 * 1e3:.4f} ms/frame   " +
              " ".join(f"{k}={ms / 20 * 1e3:.0f}" for k, (ms, c) in pr.items() if c and k.startswith("B.")))
        eh.close()
