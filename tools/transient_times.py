"""Wall time of each of the first iterations of bench.py's workload (the transient: rows bind, Newton solves)."""
import os
import sys
import time

for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine          # noqa: E402
from revs_admm_amd.synthetic import make_workload    # noqa: E402

w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
for rep in range(2):
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh,
                   mode="pdhg", feeder=w.feeder)
    ms = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.step(write_sc=False)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    print("engine", rep, "total %.2f ms" % sum(ms), [round(x, 2) for x in ms], "evaluations", e.op_iters_hist[:12], flush=True)
