#!/usr/bin/env python3
"""Operator work on harder-stressed synthetic feeders, dual Newton path and ADMM forms side
by side.  python tools/stress_diag.py [homes] [nodes] [--admm]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from revs_admm_amd.engine import AdmmEngine, OperatorOptions
from revs_admm_amd.synthetic import make_workload

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 20000
M = int(args[1]) if len(args) > 1 else 512
cases = [("newton", OperatorOptions())]
if "--admm" in sys.argv:
    cases.append(("admm", OperatorOptions(solver="admm", max_iter=4000)))
for stress in (1.15, 1.5, 2.0, 3.0, 6.0):
    for mode in ("relaxed_exact", "binary"):
        w = make_workload(n, 24, n_nodes=M, seed=0, binary_feasible=(mode == "binary"), stress=stress)
        for tag, op in cases:
            e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                           vlow=w.vlow, vhigh=w.vhigh, mode=mode, op=op)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                e.step(write_sc=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            clamped = int((e.P_est == 0).sum().item())
            ysup = int((e.yd[0] != 0).sum(dim=0).max().item())
            print(f"stress {stress} {mode:13s} {tag:6s} {dt * 1e3:8.1f} ms/30 it  paths "
                  f"{''.join(p[0] for p in e.op_path_hist)} work {e.op_iters_hist} newton "
                  f"{[h[0] for h in e.newton_hist]} clamped {clamped} max multipliers/slot {ysup}",
                  flush=True)
