#!/usr/bin/env python3
"""Binding voltage rows along the ADMM run of the bench workload.  python tools/active_rows.py [stress]"""
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
    Q, lam = e.Q.cpu().numpy(), e.s.cpu().numpy()
    Rs = (Q * lam[None, :]) @ Q.T
    sq = e.sqrt_n.cpu().numpy()[:, None]
    for k in range(1, 101):
        e.step(write_sc=False)
        if k in (2, 3, 5, 10, 20, 30, 50, 100) and e.op_path_hist[-1] == "node":
            p0, d = e.p0.cpu().numpy(), e.dnode.cpu().numpy()
            v0, v = Rs @ p0, Rs @ (p0 + d)
            hi = sq * e.vhi
            viol0 = v0 > hi * (1 + 1e-9)
            act = v >= hi * (1 - 1e-6)
            out.append((k, int(viol0.sum()), int(act.sum()), int(act.any(axis=0).sum()),
                        int((e.yv != 0).sum().item()), e.op_iters_hist[-1],
                        round(float((v0 / hi).max()), 3), round(e.residuals(1e-4)[2], 6)))
    print("paths", "".join(p[0] for p in e.op_path_hist))
    print("inner", e.op_iters_hist)
    print("stress", stress, "(iter, rows violated by p0, rows at the limit, slots, y!=0, inner its, "
          "max v0/limit, max diff):", out, flush=True)
