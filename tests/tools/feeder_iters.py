#!/usr/bin/env python3
"""The reference's own case on the GPU: 15 ADMM iterations on the 121144 feeder (1126
residences, one per node, binary homes) -- wall time and operator work per iteration.
python tests/tools/feeder_iters.py [--admm]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import golden_homes
from oracle import revs_oracle as ro
from revs_admm_amd.engine import AdmmEngine, OperatorOptions, pack_homes

z, fd = ro.load_golden(os.path.join(ROOT, "tests", "golden", "revs_121144.npz"))
R = ro.compute_Rmat_tree(fd)
nonsub, res = fd.nonsub(), fd.res()
pos = -np.ones(fd.n_nodes, np.int64)
pos[nonsub] = np.arange(len(nonsub))
Rr = R[np.ix_(pos[res], pos[res])]
oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
n = oh.LOAD.shape[0]
def run(label, op):
    e = AdmmEngine(z["tariff_shift6"], pack_homes(oh.ev, 4.8, 20.0, 0.2, 11, 23), oh.LOAD,
                   np.arange(n), Rr, kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode="binary",
                   op=op)
    for rep in range(2):                      # second run: code objects loaded
        for t in (e.P_est, e.P_sch, e.G):
            t.zero_()
        for y in e.yd:
            y.zero_()
        e._y_support = e._spec_ok = False
        e.op_cold = e._fast_cold = True
        e.op_iters_hist.clear(); e.newton_hist.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run(15)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{label}: {dt * 1e3:8.1f} ms for 15 ADMM iterations; operator evaluations / inner "
          f"iterations {sum(e.op_iters_hist)}  {e.op_iters_hist}  newton {[h[0] for h in e.newton_hist]}",
          flush=True)


run("dual Newton (default)", OperatorOptions())
if "--kadd" in sys.argv:
    for _k in (1, 2, 4, 8):
        run(f"dual Newton, kadd {_k}", OperatorOptions(newton_kadd=_k))
if "--kadd-cold" in sys.argv:      # rows admitted per Newton iteration while a slot shows many violated rows (round 5)
    for _c, _at in ((4, 2), (6, 3), (8, 4), (8, 8), (12, 6), (16, 6), (16, 12), (32, 8)):
        run(f"dual Newton, kadd 2, cold {_c} above {_at} violated", OperatorOptions(newton_kadd_cold=_c, newton_kadd_cold_at=_at))
if "--nks" in sys.argv:            # column slabs of the Gram launch (round 5, after the products moved to the matrix cores)
    for _n in (4, 6, 8, 12, 16, 24):
        run(f"dual Newton, {_n} column slabs", OperatorOptions(newton_nks=_n))
        run(f"dual Newton, {_n} column slabs", OperatorOptions(newton_nks=_n))
if "--admm" in sys.argv:
    for rv in (1.0, 25.0):
        run(f"ADMM forms, rho_v {rv}", OperatorOptions(solver="admm", rho_v_scale=rv))
