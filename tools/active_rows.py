#!/usr/bin/env python3
"""Binding voltage rows along the ADMM run of the bench workload (dual Newton path): rows at
the limit, rows with a multiplier, Newton iterations.  python tools/active_rows.py [stress ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from revs_admm_amd.engine import AdmmEngine
from revs_admm_amd.synthetic import make_workload

for stress in [float(a) for a in sys.argv[1:]] or [1.3]:
    w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=stress)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                   vlow=w.vlow, vhigh=w.vhigh, mode="pdhg")
    out = []
    for k in range(1, 101):
        e.step(write_sc=False)
        if k in (2, 3, 5, 10, 20, 30, 50, 100):
            torch.cuda.synchronize()
            v, y = e.vfull.cpu().numpy(), e.yd[0].cpu().numpy()
            act = v >= e.vhi * (1 - 1e-6)
            out.append((k, int(act.sum()), int(act.any(axis=0).sum()), int((y != 0).sum()),
                        int((y != 0).sum(axis=0).max()), e.newton_hist[-1][0], e.op_iters_hist[-1],
                        round(float(v.max() / e.vhi), 4), int((e.P_est == 0).sum().item()),
                        round(e.residuals(1e-4)[2], 6)))
    print("paths", "".join(p[0] for p in e.op_path_hist))
    print("evaluations", e.op_iters_hist)
    print("stress", stress, "(iter, rows at the limit, slots with one, y != 0, most per slot, newton "
          "iterations, evaluations, max v/limit, clamped (home, slot) pairs, max diff):", out, flush=True)
