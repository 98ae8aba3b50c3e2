"""Python-side profile of the transient as one call (cProfile around AdmmEngine.run_steps on a second engine)."""
import cProfile
import os
import pstats
import sys
import time

for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine          # noqa: E402
from revs_admm_amd.synthetic import make_workload    # noqa: E402

w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for rep in range(3):
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh,
                   mode="pdhg", feeder=w.feeder)
    torch.cuda.synchronize()
    pr = cProfile.Profile() if rep == 2 else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    e.run_steps(n)
    if pr:
        pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("engine", rep, "%d iterations: host %.3f ms, with the closing synchronize %.3f ms" % (n, (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3), flush=True)
    if pr:
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)
