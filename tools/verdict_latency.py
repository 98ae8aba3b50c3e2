"""When does the host see the verdict that the selection workgroups write from inside the
sweep's launch?  1M homes (sweep ~147 us): host time from the launch call to the sequence tag
in pinned memory, against the time to the end of the launch.  python tools/verdict_latency.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine          # noqa: E402
from revs_admm_amd.synthetic import make_workload    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = make_workload(n, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
               vlow=w.vlow, vhigh=w.vhigh, mode="pdhg")
for _ in range(35):
    e.step(write_sc=False)
torch.cuda.synchronize()
p_scr, pe_scr = torch.zeros_like(e.pnq[0]), torch.zeros_like(e.P_est)
tags = e.stats_host[1].numpy()[:, 5]
seen, done = [], []
for rep in range(30):
    tags[:] = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.replay_sweep(p_scr, pe_scr, True)              # (its selection writes tag -1 into set 1)
    t1 = time.perf_counter()
    while not (tags == -1.0).all():
        pass
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    seen.append((t2 - t0) * 1e6)
    done.append((t3 - t0) * 1e6)
print(f"launch call {1e6 * (t1 - t0):.1f} us; tag seen after {np.median(seen):.1f} us (min {np.min(seen):.1f}); "
      f"launch finished after {np.median(done):.1f} us")
