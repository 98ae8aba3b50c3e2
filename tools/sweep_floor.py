#!/usr/bin/env python3
"""Where the sweep's time goes (needs an MI355X): the streaming steady state timed on workloads
with no EV at all (loads + epilogue + stores: no solve), the default 50 % and 100 % adoption.
    python tools/sweep_floor.py [homes] [T]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from revs_admm_amd.engine import AdmmEngine
from revs_admm_amd.synthetic import make_workload

homes = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for adoption in (0.0, 0.5, 1.0):
    w = make_workload(homes, T, n_nodes=2048, seed=0, binary_feasible=False, stress=0.8, adoption=adoption)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow,
                   vhigh=w.vhigh, mode="pdhg", feeder=w.feeder)
    e.run_steps(60)
    for _ in range(1500):
        e._gemm1(e.R64T, e.pnq[2], e.v_sl)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        e.run_steps(200)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 200 * 1e6)
    st = e.status.cpu().numpy() >> 8
    print(f"adoption {adoption}: {np.round(ts, 2).tolist()} us/step, spec {e.spec_hist}, PDHG iterations mean "
          f"{st[st > 0].mean() if (st > 0).any() else 0:.1f}")
