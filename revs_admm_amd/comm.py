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
