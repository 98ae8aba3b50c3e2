#!/usr/bin/env python3
"""Per-iteration wall time and operator work along a run (needs an MI355X):
    python tools/step_times.py [--mode binary] [--stress 1.0] [--iters 140] [--dense] [--homes 100000]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from revs_admm_amd.engine import AdmmEngine, OperatorOptions
from revs_admm_amd.synthetic import make_workload

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="binary")
ap.add_argument("--stress", type=float, default=1.0)
ap.add_argument("--iters", type=int, default=140)
ap.add_argument("--homes", type=int, default=100_000)
ap.add_argument("--T", type=int, default=24)
ap.add_argument("--dense", action="store_true")
ap.add_argument("--chunk", type=int, default=1, help="run_steps chunk (1: per-iteration times)")
ap.add_argument("--spin", type=int, default=0, help="iterations run in one call before the timed chunks")
ap.add_argument("--clock-warm", type=int, default=0, help="throw-away products before the timed chunks (bench.py does 2000)")
a = ap.parse_args()
w = make_workload(a.homes, a.T, n_nodes=2048, seed=0, binary_feasible=(a.mode == "binary"), stress=a.stress)
e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow,
               vhigh=w.vhigh, mode=a.mode, feeder=None if a.dense else w.feeder,
               op=OperatorOptions(voltage="dense" if a.dense else "auto"))
import collections
acc = collections.Counter()
cnt = collections.Counter()
def wrap(name):
    f = getattr(e, name)
    def g(*x, **kw):
        t0 = time.perf_counter()
        r = f(*x, **kw)
        acc[name] += time.perf_counter() - t0
        cnt[name] += 1
        return r
    setattr(e, name, g)
for nm in ("_chain_run", "_stream_run", "step", "_chain_finish", "_spec_discard", "_operator_solve_newton", "agent_step"):
    wrap(nm)
if a.spin:
    e.run_steps(a.spin)
    torch.cuda.synchronize()
    acc.clear(); cnt.clear()
for _ in range(a.clock_warm):
    e._gemm1(e.R64T, e.pnq[2], e.v_sl)
torch.cuda.synchronize()
ts = []
k = 0
while k < a.iters:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.run_steps(a.chunk)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3 / a.chunk)
    k += a.chunk
ts = np.array(ts)
print("ms per iteration:", np.round(ts, 3).tolist())
print("newton (iters, evals, pivots):", e.newton_hist)
print("spec", e.spec_hist, "chain", e.chain_hist, "paths", "".join(p[0] for p in e.op_path_hist))
print("host time by method (ms, calls):", {k: (round(v * 1e3, 2), cnt[k]) for k, v in acc.items()})
st = e.status.cpu().numpy() >> 8
if a.mode == "pdhg":
    it = st[e.perm.argsort()] if False else st
    grp = it[: len(it) // 8 * 8].reshape(-1, 8).max(1)
    print("PDHG iterations per home: mean %.1f p50 %d p90 %d p99 %d max %d; per wavefront (max of 8): mean %.1f p50 %d p90 %d p99 %d max %d; wavefronts with 0: %.2f"
          % (it[it > 0].mean(), *np.percentile(it[it > 0], [50, 90, 99]), it.max(), grp.mean(), *np.percentile(grp, [50, 90, 99]), grp.max(), (grp == 0).mean()))
print(f"mean of last half {ts[len(ts)//2:].mean():.4f} ms, median {np.median(ts[len(ts)//2:]):.4f} ms")
