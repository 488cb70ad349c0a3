"""CPU: the N>1 path of bench.py (sequence sharding + nav-record gather) with world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rebvo_amd import shard


def test_partition_is_balanced_and_complete():
    for n, world in ((8, 8), (10, 4), (3, 2), (1, 2)):
        parts = [shard.shard_sequences(n, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


class _Nav:
    def __init__(self, frame, seq):
        self.frame, self.kn, self.klm_num, self.estimation_ok = frame, 100 + seq, 90 + seq, 1
        self.Pos, self.V, self.W = [seq, frame, 0.5], [0.1 * seq] * 3, [0.01 * frame] * 3


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = shard.shard_sequences(4, rank, world)
    navs = [[_Nav(k, s) for s in ids] for k in range(3)]
    rec = shard.nav_records(navs, rank, ids)
    out = shard.gather_records(rec, dst=0)
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks timing reduction of bench.py
    if rank == 0:
        q.put((out, float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    assert out.shape == (2, 3, 2, shard.NAV_FIELDS)
    assert sorted(out[:, 0, :, 13].ravel().tolist()) == [0, 1, 2, 3]   # every sequence reported once
    assert np.all(out[1, :, :, 14] == 1) and np.all(out[0, :, :, 14] == 0)
    assert np.allclose(out[1, 2, 0, 4:7], [2, 2, 0.5])


def test_nav_records_from_structured_log_equals_object_path():
    """bench.py reads the per-frame nav log of a 1024-sequence batch as one numpy structured array
    (EdgeHip.read_nav_log_array): the records built from it equal the per-object path, field for field."""
    import numpy as np
    from rebvo_amd import edgehip, shard
    K, B = 5, 7
    rs = np.random.RandomState(3)
    arr = np.zeros((K, B), dtype=edgehip.NAV_DTYPE)
    arr["frame"] = np.arange(K)[:, None]
    arr["kn"] = rs.randint(0, 16000, (K, B))
    arr["klm_num"] = rs.randint(0, 16000, (K, B))
    arr["estimation_ok"] = rs.randint(0, 2, (K, B))
    for f in ("Pos", "V", "W"):
        arr[f] = rs.normal(size=(K, B, 3))
    objs = (edgehip.Nav * (K * B)).from_buffer(bytearray(arr.tobytes()))
    rows = [[objs[k * B + s] for s in range(B)] for k in range(K)]
    seq_ids = list(range(100, 100 + B))
    a = shard.nav_records(arr, 3, seq_ids)
    b = shard.nav_records(rows, 3, seq_ids)
    assert a.shape == (K, B, shard.NAV_FIELDS) and np.array_equal(a, b)
    assert np.array_equal(a[..., 13], np.tile(np.arange(100, 100 + B), (K, 1))) and (a[..., 14] == 3).all()


class _FakeLog:
    """What bench.py hands the NavMover: something with EdgeHip.read_nav_log_array(first, count)."""

    def __init__(self, rank, nseq):
        self.rank, self.nseq = rank, nseq

    def read_nav_log_array(self, first, count):
        from rebvo_amd import edgehip
        arr = np.zeros((count, self.nseq), dtype=edgehip.NAV_DTYPE)
        arr["frame"] = np.arange(first, first + count)[:, None]
        arr["kn"] = 1000 * self.rank + np.arange(self.nseq)[None, :]
        arr["Pos"][:, :, 0] = self.rank
        return arr


def _mover_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, K, Wm = 3, 10, 5
    mover = shard.NavMover(world, rank, "gloo")
    log = _FakeLog(rank, B)
    blk, posted = max(1, K // 4), 0
    dist.barrier()
    for k in range(Wm, Wm + K):                     # bench.py's posting pattern (N > 1 branch)
        done = k - Wm + 1
        if done % blk == 0 or done == K:
            mover.post(log, Wm + posted, done - posted, [rank * B + s for s in range(B)])
            posted = done
    blocks = mover.finish()
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)        # the default group stays usable next to the mover's own
    dist.barrier()
    if rank == 0:
        q.put((blocks, float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_nav_mover_world4_gloo_delivers_every_step_once():
    """bench.py's N > 1 control flow over 4 ranks: blocks of steps posted while the replay goes on, gathered on a second
    process group by the mover's thread, the default group free for the timing reduction."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 4
    procs = [ctx.Process(target=_mover_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    blocks, tmax = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 4.0
    allrec = np.concatenate(blocks, axis=1)         # [world, K, B, fields]
    assert allrec.shape == (world, 10, 3, shard.NAV_FIELDS)
    for r in range(world):
        assert np.array_equal(allrec[r, :, 0, 0], np.arange(5, 15))          # every timed step once, in order
        assert np.all(allrec[r, :, :, 14] == r) and np.all(allrec[r, :, :, 4] == r)
        assert sorted(set(allrec[r, 0, :, 13].tolist())) == [3 * r, 3 * r + 1, 3 * r + 2]


class _FailingLog(_FakeLog):
    """A reader that fails on the second block of one rank (a device error under edgehip_read_nav_log)."""

    def __init__(self, rank, nseq, bad_rank):
        super().__init__(rank, nseq)
        self.bad, self.calls = rank == bad_rank, 0

    def read_nav_log_array(self, first, count):
        self.calls += 1
        if self.bad and self.calls == 2:
            raise RuntimeError("read_nav_log: injected failure")
        return super().read_nav_log_array(first, count)


def _failing_mover_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mover = shard.NavMover(world, rank, "gloo")
    log = _FailingLog(rank, 3, bad_rank=1)
    dist.barrier()
    outcome = "no error"
    try:
        for b in range(8):                               # more posts than the queue holds: a dead worker must not block them
            mover.post(log, 5 * b, 5, [0, 1, 2], timeout=30)
        mover.finish(timeout=30)
    except Exception as e:
        outcome = f"{type(e).__name__}: {e}"
    dist.barrier()                                        # every rank got out: nobody sits in a gather or a queue.put
    q.put((rank, outcome, len(mover.blocks)))
    dist.barrier()
    dist.destroy_process_group()


def test_nav_mover_reader_failure_on_one_rank_stops_all_ranks_without_hanging():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_failing_mover_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (o, n)) for r, o, n in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "injected failure" in res[1][0]                # the failing rank reports its own error
    assert "another rank failed" in res[0][0]             # the other one is told, instead of waiting in dist.gather
    assert res[0][1] == 1                                 # the block before the failure arrived on rank 0
