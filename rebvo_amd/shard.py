"""Multi-GPU layout of the path: independent sequences shard across ranks, one process per GPU.

The edge pipeline has no intra-sequence parallelism across GPUs (frame k needs frame k-1's depth map,
SURVEY.md §8e), so ranks never exchange data on the data path.  The only collective is the gather of
the small per-frame nav records to rank 0 (RCCL over xGMI with backend "nccl"; "gloo" in CPU tests).
"""
import numpy as np

NAV_FIELDS = 16  # frame, kn, klm_num, estimation_ok, Pos[3], V[3], W[3], seq, rank, pad


def shard_sequences(n_total, rank, world):
    """Contiguous, balanced partition of sequence ids [0, n_total) — rank r gets ids[r]."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def nav_records(navs, rank, seq_ids):
    """navs: [steps][nseq] objects with frame/kn/klm_num/estimation_ok/Pos/V/W -> float64 [steps, nseq, 16]."""
    rec = np.zeros((len(navs), len(seq_ids), NAV_FIELDS))
    if isinstance(navs, np.ndarray) and navs.dtype.names:   # structured array (EdgeHip.read_nav_log_array): no Python loop
        for j, f in enumerate(("frame", "kn", "klm_num", "estimation_ok")):
            rec[:, :, j] = navs[f]
        rec[:, :, 4:7], rec[:, :, 7:10], rec[:, :, 10:13] = navs["Pos"], navs["V"], navs["W"]
        rec[:, :, 13] = np.asarray(seq_ids)[None, :]
        rec[:, :, 14] = rank
        return rec
    for k, row in enumerate(navs):
        for s, n in enumerate(row):
            rec[k, s, :4] = (n.frame, n.kn, n.klm_num, n.estimation_ok)
            rec[k, s, 4:7] = n.Pos[:]
            rec[k, s, 7:10] = n.V[:]
            rec[k, s, 10:13] = n.W[:]
            rec[k, s, 13] = seq_ids[s]
            rec[k, s, 14] = rank
    return rec


def nav_records_device(raw, rank, seq_ids):
    """The same [steps, nseq, 16] float64 records as nav_records(), formed ON THE DEVICE from the raw edgehip_nav structs that
    edgehip_read_nav_log_device copied out of the context's log (raw: uint8 tensor [steps, nseq, sizeof(edgehip_nav)]): column
    picks on two reinterpreting views, no host round trip — the tensor that comes out is what RCCL sends."""
    import torch
    from .edgehip import NAV_DTYPE
    steps, nseq, nbytes = raw.shape
    assert nbytes == NAV_DTYPE.itemsize and nbytes % 8 == 0
    f64 = raw.view(torch.float64)             # [steps, nseq, nbytes / 8]
    i32 = raw.view(torch.int32)               # [steps, nseq, nbytes / 4]
    off = {name: NAV_DTYPE.fields[name][1] for name in NAV_DTYPE.names}
    rec = torch.zeros((steps, nseq, NAV_FIELDS), dtype=torch.float64, device=raw.device)
    for j, f in enumerate(("frame", "kn", "klm_num", "estimation_ok")):
        rec[:, :, j] = i32[:, :, off[f] // 4].to(torch.float64)
    for j, f in ((4, "Pos"), (7, "V"), (10, "W")):
        rec[:, :, j:j + 3] = f64[:, :, off[f] // 8: off[f] // 8 + 3]
    rec[:, :, 13] = torch.as_tensor(list(seq_ids), dtype=torch.float64, device=raw.device)[None, :]
    rec[:, :, 14] = float(rank)
    return rec


def gather_records(rec, dst=0, group=None):
    """Gather equally-shaped record tensors to `dst` (torch.distributed must be initialised).

    Returns [world, ...] on dst, None elsewhere.  Tensors live on the GPU for nccl(=RCCL), on the host for gloo."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank()
    t = rec if isinstance(rec, torch.Tensor) else torch.as_tensor(rec)
    if dist.get_backend(group) == "nccl" and not t.is_cuda:
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, out, dst=dst, group=group)
    return torch.stack(out).cpu().numpy() if rank == dst else None


class NavMover:
    """Ships the per-frame nav records of a replay to rank 0 while the replay goes on (the
    payload is ~0.5 KB per frame, i.e. pure latency, so it must stay off the critical path).

    A worker thread takes blocks of steps, reads them out of the context's device-side nav log (a small copy ordered
    after the frames it covers; the thread waits for it, the thread that enqueues frames does not), and gathers them to
    rank 0 over its OWN process group / communicator (RCCL for backend "nccl": its own stream, never interleaved with the
    collectives of the main thread).  At most two blocks are in flight (double buffering): `post` only blocks when the
    transport falls two blocks behind.  Every rank must post the same blocks in the same order."""

    def __init__(self, world, rank, backend=None, dst=0, device=None):
        import queue
        import threading
        import torch.distributed as dist
        self.rank, self.dst, self.device = rank, dst, device
        self.group = dist.new_group(ranks=list(range(world)), backend=backend)   # collective: all ranks call it
        self.blocks = []                      # on dst: one [world, steps, nseq, NAV_FIELDS] array per posted block
        self.error = None
        self.device_path_blocks = 0           # blocks whose records went from the device log to the communicator without touching the host
        self._q = queue.Queue(maxsize=2)
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _put(self, item, timeout):
        """Blocking put that keeps looking at the worker: a dead or failed worker never drains the queue."""
        import queue
        import time
        t_end = None if timeout is None else time.monotonic() + timeout
        while True:
            if self.error is not None:
                raise self.error
            if not self._t.is_alive():
                raise RuntimeError("nav mover thread is gone")
            try:
                self._q.put(item, timeout=0.2)
                return
            except queue.Full:
                if t_end is not None and time.monotonic() > t_end:
                    raise TimeoutError(f"nav gather is more than two blocks behind after {timeout} s")

    def post(self, log_reader, first, count, seq_ids, timeout=None):
        """log_reader.read_nav_log_array(first, count) -> structured array [count, nseq] (EdgeHip does)."""
        self._put((log_reader, first, count, list(seq_ids)), timeout)

    def _run(self):
        import torch
        import torch.distributed as dist
        if self.device is not None:           # the current device is per thread
            torch.cuda.set_device(self.device)
        failed = False
        while True:
            item = self._q.get()
            if item is None:
                return
            if failed:
                continue                      # keep draining: a post() blocked on the full queue must wake up and see the error
            rec = None
            try:
                reader, first, count, seq_ids = item
                if self.device is not None and hasattr(reader, "read_nav_log_device"):
                    # RCCL: HBM -> wire.  The raw records are copied device-to-device out of the context's log (on the log's own
                    # stream, complete on return), the 16 fields are picked on the device, and that tensor is what the gather sends.
                    from .edgehip import NAV_DTYPE
                    raw = torch.empty((count, len(seq_ids), NAV_DTYPE.itemsize), dtype=torch.uint8, device=torch.device("cuda", self.device))
                    reader.read_nav_log_device(first, count, raw.data_ptr())
                    rec = nav_records_device(raw, self.rank, seq_ids)
                    self.device_path_blocks += 1
                else:
                    rec = nav_records(reader.read_nav_log_array(first, count), self.rank, seq_ids)
            except Exception as e:            # surfaces in post() / finish(); the replay itself must not be torn down by the transport
                self.error = e
            try:
                # Every rank reaches the same collectives in the same order whatever happened locally: a status word first,
                # so that one rank's failure stops the transport on all of them instead of leaving the others in the gather.
                ok = torch.tensor([0 if rec is None else 1], dtype=torch.int32)
                if dist.get_backend(self.group) == "nccl":
                    ok = ok.cuda()
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
                if int(ok.item()) == 0:
                    if self.error is None:
                        self.error = RuntimeError("nav gather stopped: another rank failed to read its records")
                    failed = True
                    continue
                out = gather_records(rec, dst=self.dst, group=self.group)
                if out is not None:
                    self.blocks.append(out)
            except Exception as e:
                if self.error is None:
                    self.error = e
                failed = True

    def finish(self, timeout=None):
        """Waits until every posted block has arrived; returns the blocks (on dst) in posting order.  With a timeout
        (seconds) a transport that does not come back raises TimeoutError instead of blocking the caller for ever (the
        worker is a daemon thread)."""
        import time
        t0 = time.monotonic()
        try:
            self._put(None, timeout)
        except Exception:
            try:
                self._q.put_nowait(None)      # let a draining worker end
            except Exception:
                pass
            if self.error is not None:
                raise self.error
            raise
        self._t.join(None if timeout is None else max(0.0, timeout - (time.monotonic() - t0)))
        if self._t.is_alive():
            raise TimeoutError(f"nav gather still running after {timeout} s")
        if self.error is not None:
            raise self.error
        return self.blocks
