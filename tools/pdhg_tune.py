"""PDHG stopping rule against accuracy and passes: one sweep from a mid-run state compared with the
oracle's exact relaxed optimum, and the closed loop's late-run diff against the oracle's.

    python tools/pdhg_tune.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import f32, oracle_homes                      # noqa: E402
from oracle import revs_oracle as ro                       # noqa: E402
from revs_admm_amd.engine import AdmmEngine                # noqa: E402
from revs_admm_amd.synthetic import make_workload          # noqa: E402


def engine(w, **pd):
    return AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow,
                      vhigh=w.vhigh, mode="pdhg", feeder=w.feeder, pdhg=pd or None)


w = make_workload(20000, 24, n_nodes=512, seed=0, binary_feasible=False, stress=1.0)
w.load, w.cost = f32(w.load), f32(w.cost)
oh = oracle_homes(w)
base = engine(w)
base.run_steps(40)
pe, ps, gm = base.get_state()
pe64, ps64, gm64 = (a.astype(np.float64) for a in (pe, ps, gm))
p_ref, *_ = ro.home_solve_relaxed(w.cost, oh, pe64, ps64, gm64, w.kappa)
print("one sweep from the state after 40 iterations (20 000 residences): |S - exact optimum| and passes")
for tol, polish in ((1e-6, 0), (1e-7, 0), (1e-6, 1), (1e-5, 1), (1e-4, 1), (1e-3, 1)):
    e = engine(w, tol=tol, polish=polish)
    e.set_state(pe, ps, gm)
    if e.pdhg_dual is not None:
        e.pdhg_dual.copy_(base.pdhg_dual)
    e.P_est_new.copy_(base.P_est_new)
    e.agent_step(write_sc=True)
    S = e.S.cpu().numpy()[e.inv_perm]
    st = e.status.cpu().numpy()[e.inv_perm]
    it = (st >> 8)[oh.ev]
    err = np.abs(S - p_ref).max(axis=1)
    print(f"  tol {tol:g} polish {polish}: max {err.max():.2e} kW, 99.9 % {np.quantile(err, 0.999):.2e}, mean {err.mean():.2e}; "
          f"passes mean {it.mean():.1f} max {it.max()}")

w2 = make_workload(600, 24, n_nodes=60, seed=21, binary_feasible=False, stress=1.02)
w2.load, w2.cost = f32(w2.load), f32(w2.cost)
iters = 150
d_ref, *_ = ro.solve_ADMM(oracle_homes(w2), w2.Rn, w2.node_of, w2.cost, w2.kappa, iters, w2.vset, w2.vlow,
                          w2.vhigh, mode="relaxed", util_method="dual")
print("closed loop, 600 x 24, 150 iterations: late-run |diff - oracle| / max diff (iterations 75..150)")
for tol, polish in ((1e-6, 0), (1e-6, 1), (1e-5, 1), (1e-4, 1), (1e-3, 1)):
    e = engine(w2, tol=tol, polish=polish)
    d = e.run(iters)
    late = slice(iters // 2, iters)
    print(f"  tol {tol:g} polish {polish}: {np.abs(d[late] - d_ref[late]).max() / d_ref[late].max():.4f} "
          f"(whole run {np.abs(d - d_ref).max():.2e}); first 10 iterations {np.abs(d[:10] - d_ref[:10]).max():.2e}")
