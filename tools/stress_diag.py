#!/usr/bin/env python3
"""Operator convergence on harder-stressed synthetic feeders.  python tools/stress_diag.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from revs_admm_amd.engine import AdmmEngine, OperatorOptions
from revs_admm_amd.synthetic import make_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
for stress in (1.15, 1.5, 2.0, 3.0):
    w = make_workload(n, 24, n_nodes=M, seed=0, binary_feasible=False, stress=stress)
    for tag, op in (("default", OperatorOptions(max_iter=4000)),):
        e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                       vlow=w.vlow, vhigh=w.vhigh, mode="relaxed_exact", op=op)
        for _ in range(30):
            e.step(write_sc=False)
        clamped = int((e.P_est == 0).sum().item())
        print(f"stress {stress} {tag:13s} paths {''.join(p[0] for p in e.op_path_hist)} iters "
              f"{e.op_iters_hist} scales {getattr(e, 'rho_scales', None)} clamped {clamped} "
              f"rho_v/b now {e.rho_v.max().item() * e.smax ** 2 / e.kappa:.3g}/"
              f"{e.rho_b.max().item() / e.kappa:.3g}", flush=True)
