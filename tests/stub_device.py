"""TEST INFRASTRUCTURE — a device stand-in for bench.py's multi-rank CONTROL FLOW on a box without a GPU.

`BENCH_STUB_DEVICE=1 python bench.py --gpus N` (tests/test_bench_launcher_cpu.py) walks everything an N-GPU run does around the
kernels — the self-launch of N ranks, the process group, barriers, shard.NavMover's block gather to rank 0, the max-over-ranks
reduction, the ONE line on rank 0's stdout — with this class in the place of rebvo_amd.edgehip.EdgeHip.  It does no image work:
process_frame() writes a deterministic record per (rank, sequence, frame) into a host-side log.  The line such a run prints says
`"data": "stub"` and `"invalid_as_measurement": true`; nothing under rebvo_amd/ imports this file, and bench.py takes it only
under that environment switch.
"""
import os

import numpy as np

from rebvo_amd import edgehip as _eh


def stub_value(rank, seq, frame, j):
    """What the stub logs as Pos[j] of (rank, seq, frame): exactly representable, so the gathered records can be checked with ==."""
    return float(rank) * 4096.0 + float(seq) + float(frame) / 1024.0 + float(j) / 8192.0


class _Nav:
    __slots__ = ("kn", "estimation_ok", "minimizer_evals", "klm_num", "frame")


class EdgeHip:
    """The methods of rebvo_amd.edgehip.EdgeHip that bench.py's full path calls, on a host-side log."""

    def __init__(self, params, nseq=1, nslots=3, device=0):
        self.nseq, self.nslots, self.rank = nseq, nslots, device
        self._frame = 0
        self._slot = -1
        self._log = {}

    def next_slot(self):
        return (self._slot + 1) % self.nslots

    def cur_slot(self):
        return self._slot

    def bind_rgb_indexed(self, slot, pool_ptr, pool_frames, idx):
        assert len(idx) == self.nseq

    def process_frame(self, t):
        k = self._frame
        if k == 2 and os.environ.get("BENCH_STUB_FAIL_RANK") == str(self.rank):   # a rank that dies mid-run (launcher test)
            raise RuntimeError("stub device: injected failure")
        rec = np.zeros(self.nseq, dtype=_eh.NAV_DTYPE)
        s = np.arange(self.nseq)
        rec["frame"], rec["kn"], rec["klm_num"], rec["estimation_ok"], rec["minimizer_evals"] = k, 12000 + s % 7, 9000 + s % 5, 1, 12
        for j in range(3):
            rec["Pos"][:, j] = self.rank * 4096.0 + s + k / 1024.0 + j / 8192.0
        rec["t"] = t
        self._log[k] = rec
        self._slot = self.next_slot()
        self._frame += 1

    def set_nav_log(self, length):
        self._log.clear()

    def read_nav_log_array(self, first, count):
        if first < 0 or first + count > self._frame:
            raise RuntimeError(f"stub nav log: frames {first}..{first + count - 1} not enqueued")
        return np.stack([self._log[first + k] for k in range(count)])

    def read_nav(self):
        out = []
        for r in self._log[self._frame - 1]:
            n = _Nav()
            n.kn, n.estimation_ok, n.minimizer_evals, n.klm_num, n.frame = int(r["kn"]), int(r["estimation_ok"]), int(r["minimizer_evals"]), int(r["klm_num"]), int(r["frame"])
            out.append(n)
        return out

    def get_kn(self, slot):
        return [12000] * self.nseq

    def sync(self):
        pass

    def close(self):
        pass

    def profile_enable(self, on):
        pass

    def profile_select(self, groups):
        pass

    def profile_read(self):
        return {}
