"""GPU: shard.NavMover against a real context — its worker thread reads blocks of the device-side nav log
(edgehip_read_nav_log) WHILE the main thread keeps enqueueing frames on the same context, which is what bench.py does for
N > 1.  Once over a one-rank gloo group (the transport is not the point) and once over "nccl" = RCCL with world_size 1 on the
one GPU a test box has: device tensors, a second communicator made by dist.new_group and driven from the worker thread after
torch.cuda.set_device, the status word read back with .item() on that thread — the branch an 8-GPU run takes, minus the
peers.  Run in a child process so that the process group does not leak into the other tests."""
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _run(port, q, backend="gloo"):
    import os
    import torch
    import torch.distributed as dist
    from rebvo_amd import edgehip, shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)
    w, h, n, K, blk = 376, 240, 3, 24, 4
    frames = [f for f, _, _ in synth.billboard_sequence(w, h, 8)]
    eh = edgehip.EdgeHip(edgehip.euroc_params(w, h), nseq=n, nslots=3)
    eh.set_nav_log(K)
    mover = shard.NavMover(1, 0, backend, device=0 if backend == "nccl" else None)
    posted = 0
    for k in range(K):
        eh.upload_rgb(eh.next_slot(), np.stack([frames[(k + s) % 8] for s in range(n)]))
        eh.process_frame(0.05 * k)                       # no synchronisation on this thread
        if (k + 1) % blk == 0:
            mover.post(eh, posted, k + 1 - posted, list(range(n)))
            posted = k + 1
    blocks = mover.finish()
    eh.sync()
    direct = shard.nav_records(eh.read_nav_log_array(0, K), 0, list(range(n)))
    got = np.concatenate([b[0] for b in blocks], axis=0)
    q.put((got.shape, bool(np.array_equal(got, direct)), int(direct[-1, 0, 0]), float(np.abs(direct[:, :, 7:10]).max())))
    eh.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_nav_mover_reads_the_log_while_frames_are_enqueued(backend):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_run, args=(port, q, backend))
    p.start()
    shape, same, last_frame, vmax = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert shape == (24, 3, 16) and same
    assert last_frame == 23 and vmax > 0
