#!/usr/bin/env python3
"""BASELINE config 3 on the GPU: the 121144 feeder, ALL communities, 90 % adoption, 4.8 kW,
T = 96 (the hourly base load and tariff held over four 15-minute slots, charging window
11:00-23:00 -> slots 44..92; the reference's slot arithmetic unchanged).  Prints the wall
time of 15 ADMM iterations per home mode, the operator's work per iteration and how many
voltage rows end up with a nonzero multiplier per slot; with --oracle the first iterations
of the relaxed run are compared with the CPU oracle.
python tests/tools/feeder_config3.py [--oracle] [--adopt 90] [--T 96]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from oracle import revs_oracle as ro
from revs_admm_amd.engine import AdmmEngine, OperatorOptions, pack_homes

ap = argparse.ArgumentParser()
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--adopt", type=int, default=90)
ap.add_argument("--T", type=int, default=96)
ap.add_argument("--iters", type=int, default=15)
ap.add_argument("--nks", type=int, default=0, help="column slabs of the Gram launch (0: the engine's choice)")
args = ap.parse_args()

z, fd = ro.load_golden(os.path.join(ROOT, "tests", "golden", "revs_121144.npz"))
R = ro.compute_Rmat_tree(fd)
nonsub, res = fd.nonsub(), fd.res()
pos = -np.ones(fd.n_nodes, np.int64)
pos[nonsub] = np.arange(len(nonsub))
Rr = R[np.ix_(pos[res], pos[res])]
rep = args.T // 24
assert rep * 24 == args.T
res_ids = z["res_id"]
n = len(res_ids)
np.random.seed(1234)                                   # revs_fixture.py:174-177, all communities
ev_homes = np.random.choice(res_ids, int(args.adopt * 1e-2 * n), replace=False)
idx = {h: i for i, h in enumerate(res_ids)}
ev = np.zeros(n, bool)
ev[[idx[h] for h in ev_homes]] = True
LOAD = np.repeat(z["LOAD"], rep, axis=1).astype(np.float32)
cost = np.repeat(z["tariff_shift6"], rep).astype(np.float32)
start, end = 11 * rep, 23 * rep
print(f"{n} residences, {ev.sum()} EVs, T = {args.T}, window [{start}, {end})", flush=True)


def run(mode):
    e = AdmmEngine(cost, pack_homes(ev, 4.8, 20.0, 0.2, start, end), LOAD, np.arange(n), Rr,
                   kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode=mode, op=OperatorOptions(newton_nks=args.nks))
    for r in range(2):
        for t in (e.P_est, e.P_sch, e.G):
            t.zero_()
        for y in e.yd:
            y.zero_()
        e._y_support = e._spec_ok = False
        e.op_cold = e._fast_cold = True
        e.op_iters_hist.clear(); e.newton_hist.clear(); e.op_path_hist.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = e.run(args.iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    y = e.yd[0].cpu().numpy()
    nz = (y != 0).sum(0) if y.shape[0] == n else (y != 0).sum(1)
    print(f"{mode:14s}: {dt * 1e3:8.1f} ms for {args.iters} ADMM iterations; evaluations "
          f"{e.op_iters_hist}; paths {sorted(set(e.op_path_hist))}; newton "
          f"{[h[0] for h in e.newton_hist]}; multipliers per slot max {nz.max()} "
          f"(slots with any: {(nz > 0).sum()}); max diff {d[-1].max():.3e}", flush=True)
    return e, d


for mode in ("binary", "pdhg"):
    e, d = run(mode)

if args.oracle:
    k = 3
    e, d = run("relaxed_exact")
    oh = ro.Homes.uniform(LOAD, ev, 4.8, 20.0, 0.2, start, end)
    t0 = time.perf_counter()
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oh, Rr, np.arange(n), cost, 5.0, k, 1.03, 0.95,
                                               1.05, mode="relaxed", util_eps=1e-10)
    print(f"oracle: {time.perf_counter() - t0:.1f} s for {k} iterations", flush=True)
    print("diff trajectory max abs error", np.abs(d[:k] - d_ref).max(), "scale", d_ref.max())
