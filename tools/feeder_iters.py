#!/usr/bin/env python3
"""Operator inner-iteration counts on the reference's 121144 feeder (binary homes, 15 ADMM
iterations) for a few initial rho scales.  python tools/feeder_iters.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
for rv in (0.1, 1.0, 5.0, 25.0, 100.0):
    for rb in (0.1, 1.0):
        op = OperatorOptions(rho_v_scale=rv, rho_b_scale=rb)
        e = AdmmEngine(z["tariff_shift6"], pack_homes(oh.ev, 4.8, 20.0, 0.2, 11, 23), oh.LOAD,
                       np.arange(n), Rr, kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode="binary",
                       op=op)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d = e.run(15)
        torch.cuda.synchronize()
        print(f"rho_v {rv:6.1f} rho_b {rb:4.1f}: {time.perf_counter() - t0:6.2f} s  inner its "
              f"{sum(e.op_iters_hist):6d}  {e.op_iters_hist}", flush=True)
