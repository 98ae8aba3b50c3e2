"""Host-side transports for the library's communicator (revs_comm_create_hook, include/revs_admm.h).

The product's collective is RCCL over xGMI (revs_comm_create: one rank per GPU).  Where ranks have
no RCCL path between them -- two ranks sharing one device (RCCL refuses that), a CPU-side process
group -- the same native loops run over a caller-supplied all-reduce: the library stages the node
sums through pinned host memory and calls back.  Used by bench.py --share-gpu and the two-process
tests of the sharded streaming loop (tests/test_gpu_sharded.py)."""
from __future__ import annotations

import numpy as np
import torch

from ._lib import HOST_ALLREDUCE_FN


def group_allreduce_hook(group):
    """fn(host_array, op): in-place all-reduce over a torch.distributed group with a CPU backend."""
    RO = torch.distributed.ReduceOp
    ops = {0: RO.SUM, 2: RO.MAX, 3: RO.MIN}

    def fn(a, op):
        torch.distributed.all_reduce(torch.from_numpy(a), op=ops[op], group=group)
    return fn


def host_hook(fn):
    """Wrap `fn(host_array, op)` as a revs_host_allreduce_fn (keep the returned object alive as
    long as the communicator)."""
    def cb(_ctx, buf, count, op):
        try:
            fn(np.ctypeslib.as_array(buf, shape=(int(count),)), int(op))
            return 0
        except Exception:        # never unwind through the C frames
            import traceback
            traceback.print_exc()
            return 1
    return HOST_ALLREDUCE_FN(cb)


class LocalRanks:
    """N logical ranks inside ONE process (one thread each), for rehearsing the sharded native loops on a single
    device beyond the two processes RCCL-less boxes allow: `LocalRanks(8).rank(r)` is handed to `AdmmEngine(group=...)`
    by thread r, and every all-reduce of that engine -- the library's hook communicator (revs_comm_create_hook) inside
    the native loops included -- becomes a barrier-synchronised in-memory reduction: every rank deposits its buffer,
    all wait, every rank reduces the N buffers in rank order (the same order on every rank: identical bits), all wait
    again before the buffers are reused.  A rank that issues another collective than its peers (other size, or none)
    breaks the barrier after `timeout` seconds instead of hanging.  Used by tests/test_gpu_sharded.py (world size 8)."""

    def __init__(self, size, timeout=120.0):
        import threading
        self.size = int(size)
        self._bar = threading.Barrier(self.size, timeout=timeout)
        self._slots = [None] * self.size
        self.calls = [[] for _ in range(self.size)]      # element counts of every collective, per rank

    def rank(self, r):
        return _LocalRank(self, int(r))

    def abort(self):
        self._bar.abort()


class _LocalRank:
    def __init__(self, world, r):
        self.world, self.rank_id, self.size = world, r, world.size

    def allreduce_host(self, a, op=0):
        """In place on a contiguous float64 numpy array; op 0 sum, 2 max, 3 min (revs_comm_allreduce_f64's codes)."""
        w = self.world
        w.calls[self.rank_id].append(int(a.size))
        w._slots[self.rank_id] = a
        w._bar.wait()
        if any(s.size != a.size for s in w._slots):
            w._bar.abort()
            raise RuntimeError(f"LocalRanks: rank {self.rank_id} reduces {a.size} elements, its peers "
                               f"{[s.size for s in w._slots]}")
        fn = {0: np.add, 2: np.maximum, 3: np.minimum}[int(op)]
        out = w._slots[0].copy()
        for s in w._slots[1:]:
            fn(out, s, out=out)
        w._bar.wait()                    # every rank has read every buffer
        a[...] = out
        w._bar.wait()                    # ... and written its own: the slots may be reused
        return a
