"""Wall time of each of the first iterations (the transient before the steady state): python tools/transient_times.py"""
import os, sys, time
for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine
from revs_admm_amd.synthetic import make_workload
w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
for rep in range(2):
    t0 = time.perf_counter()
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode="pdhg", feeder=w.feeder)
    torch.cuda.synchronize()
    print(f"engine set up in {(time.perf_counter() - t0) * 1e3:.1f} ms")
    ts = []
    for k in range(40):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e.step(write_sc=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t1) * 1e3)
    print("ms per iteration:", " ".join(f"{x:.3f}" for x in ts))
    print("operator evaluations:", e.op_iters_hist[:40], "total", round(sum(ts), 2), "ms")
