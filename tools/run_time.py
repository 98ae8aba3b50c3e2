"""Wall time of AdmmEngine.run(K) -- what lpsolver.solve_ADMM calls -- at the bench size: the whole
loop from a cold state, the per-iteration diff of every residence fetched at the end.
python tools/run_time.py [iterations] [homes]"""
import os, sys, time
for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine
from revs_admm_amd.synthetic import make_workload
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
w = make_workload(n, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
for rep in range(3):
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow,
                   vhigh=w.vhigh, mode="pdhg", feeder=w.feeder)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d = e.run(K)
    t1 = time.perf_counter()
    print(f"run({K}) on {n} residences: {(t1 - t0) * 1e3:.1f} ms, native bursts {len(e.stream_calls)}, "
          f"max diff of the last iteration {d[-1].max():.3e}", flush=True)
