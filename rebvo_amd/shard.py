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


def gather_records(rec, dst=0):
    """Gather equally-shaped record tensors to `dst` (torch.distributed must be initialised).

    Returns [world, ...] on dst, None elsewhere.  Tensors live on the GPU for nccl(=RCCL), on the host for gloo."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.as_tensor(rec)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, out, dst=dst)
    return torch.stack(out).cpu().numpy() if rank == dst else None
