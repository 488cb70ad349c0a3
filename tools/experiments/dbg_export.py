import numpy as np, sys
sys.path.insert(0, "/root/repo")
from rebvo_amd import edgehip, synth
W,H=376,240
T0,DT=1.0,0.05
def tri(k, n):
    p = 2 * (n - 1); k %= p
    return k if k < n else p - k
def batched():
    n_obj, pool = 5, 6
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=21)]
    eh = edgehip.EdgeHip(edgehip.euroc_params(W, H), nseq=n_obj, nslots=3, device=0)
    for k in range(4):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[tri(k + i, pool)] for i in range(n_obj)]))
        eh.process_frame(np.full(n_obj, T0 + DT * k))
    for slot in ((eh.cur_slot() + 2) % 3, eh.cur_slot()):
        seqs = [4, 0, 2]
        for registered in ((), (0, 2)):
            got = eh.download_keylines_batch(slot, seqs, registered=registered)
    eh.close()
def exp():
    n_obj, pool, n_fr = 5, 6, 9
    frames = [f for f, _, _ in synth.billboard_sequence(W, H, pool, seed=23)]
    p = edgehip.euroc_params(W, H)
    eh, ref = edgehip.EdgeHip(p, nseq=n_obj, nslots=3, device=0), edgehip.EdgeHip(p, nseq=n_obj, nslots=3, device=0)
    seqs = [4, 0, 2]
    want, tickets, got = {}, {}, {}
    def collect(k):
        kns = [len(want[k][j]) for j in range(len(seqs))]
        f = eh.export_fetch(tickets.pop(k), kns, registered=(k % 2 == 0))
        got[k] = eh.export_wait(f)
    for k in range(n_fr):
        batch = np.stack([frames[tri(k + i, pool)] for i in range(n_obj)])
        for e in (eh, ref):
            e.upload_rgb(e.next_slot(), batch)
            e.process_frame(np.full(n_obj, T0 + DT * k))
        if k >= 1:
            tickets[k] = eh.export_keylines(seqs)
            want[k] = [ref.download_keylines(s_, (ref.cur_slot() + 2) % 3, want_mask=False)[0] for s_ in seqs]
        if k >= 3:
            collect(k - 2)
    for k in list(tickets):
        collect(k)
    for k in got:
        for j,(a, b) in enumerate(zip(got[k], want[k])):
            av=a.view(np.uint8).reshape(len(a),168); bv=b.view(np.uint8).reshape(len(b),168)
            bad=(av!=bv).any(axis=1)
            fields=[n for n in a.dtype.names if not np.array_equal(a[n],b[n])]
            print(k,j,len(a),int(bad.sum()), fields[:8], np.flatnonzero(bad)[:5])
    for e in (eh, ref): e.close()
if len(sys.argv)>1: batched()
exp()
