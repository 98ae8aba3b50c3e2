"""Per-iteration wall time of the first ADMM iterations (the transient) at the bench size, with
the operator's evaluations / Newton iterations of each.  python tools/transient_times.py [homes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine          # noqa: E402
from revs_admm_amd.synthetic import make_workload    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
w = make_workload(n, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
for rep in range(2):
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                   vlow=w.vlow, vhigh=w.vhigh, mode="pdhg")
    torch.cuda.synchronize()
    out = []
    for k in range(14):
        t0 = time.perf_counter()
        e.step(write_sc=False)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    print("engine", rep, " ".join(f"{x:.3f}" for x in out), flush=True)
    print("   evaluations", e.op_iters_hist, "newton", [h[0] for h in e.newton_hist], "model calls", e.model_calls)
