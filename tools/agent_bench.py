"""Micro-benchmark of revs_agent_step alone (MI355X): launch time of the home sweep for
several PDHG settings from a mid-run ADMM state; max_iter = check = 1 gives (nearly) the kernel's
load/epilogue/store floor.   python tools/agent_bench.py [homes] [T]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine          # noqa: E402
from revs_admm_amd.synthetic import make_workload    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
w = make_workload(n, T, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
for mode in ("pdhg", "relaxed_exact", "binary"):
    variants = [dict()] if mode != "pdhg" else [dict(), dict(max_iter=1, check=1), dict(check=1), dict(check=2),
                                                dict(check=8)]
    for var in variants:
        e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                       vlow=w.vlow, vhigh=w.vhigh, mode=mode)
        for _ in range(35):
            e.step(write_sc=False)
        for k, v in var.items():
            setattr(e.pdhg, k, v)
        snap = [t.clone() for t in (e.P_sch, e.G)] + ([e.pdhg_dual.clone()] if e.pdhg_dual is not None else [])
        ts = []
        for rep in range(40):
            for t, c in zip((e.P_sch, e.G, e.pdhg_dual), snap):
                t.copy_(c)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            e.agent_step(write_sc=False)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        it = (e.status.cpu().numpy() >> 8).mean()
        print(f"{mode:14s} {str(var):22s} median {np.median(ts):7.2f} us  min {np.min(ts):7.2f} us  "
              f"pdhg iters/home {it:.1f}  -> {n * T * 4 * 7.5 / np.median(ts) / 1e6:.2f} TB/s", flush=True)
