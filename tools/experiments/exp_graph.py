"""Single-sequence (and small-batch) frame latency with and without whole-frame HIP graphs.  usage: exp_graph.py"""
import os, subprocess, sys
code = '''
import sys, time
sys.path.insert(0, ".")
import numpy as np
from rebvo_amd import edgehip, synth
B = int(sys.argv[1])
frames = [f for f, _, _ in synth.billboard_sequence(752, 480, 30)]
eh = edgehip.EdgeHip(edgehip.euroc_params(), nseq=B, nslots=3)
batches = [np.ascontiguousarray(np.stack([f] * B)) for f in frames]
for rep in range(3):
    eh.reset(); eh.sync()
    lat = []
    for k, b in enumerate(batches):
        eh.upload_rgb(eh.next_slot(), b)
        eh.sync()
        t0 = time.perf_counter()
        eh.process_frame(0.05 * k)
        eh.sync()
        lat.append(time.perf_counter() - t0)
print("B=%d graph=%s: median frame latency %.3f ms (frames 6..)" % (B, __import__("os").environ.get("EDGEHIP_GRAPH"), 1e3 * float(np.median(lat[6:]))))
'''
for B in (1, 8, 64):
    for g in ("0", "1"):
        env = dict(os.environ, EDGEHIP_GRAPH=g)
        print(subprocess.run([sys.executable, "-c", code, str(B)], env=env, capture_output=True, text=True).stdout.strip())
