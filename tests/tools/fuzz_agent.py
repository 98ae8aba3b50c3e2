#!/usr/bin/env python3
"""Differential fuzz of revs_agent_step against the oracle over seeds, horizons and modes.
    python tests/tools/fuzz_agent.py [n_seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle import revs_oracle as ro
from revs_admm_amd import _lib
import test_gpu_agent as tg

lib = _lib.load()
nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 10
worst = {}
for seed in range(nseed):
    for T in (5, 24, 31, 48, 96, 160):
        for mode in ("binary", "relaxed_exact", "pdhg"):
            w, oh = tg._prep(1537, T, seed=1000 * seed + T, binary_feasible=(mode == "binary"))
            pe_old, pe_new, ps, gm = tg._state(w, seed + T, scale=1.0 + seed % 3)
            r = tg._run_agent(lib, w, pe_old, pe_new, ps, gm, mode)
            solve = ro.home_solve_binary if mode == "binary" else ro.home_solve_relaxed
            p, s, g, st = solve(w.cost, oh, pe_old, ps, gm, w.kappa)
            assert ((r["status"] & 0xFF) == st).all(), (seed, T, mode)
            ok = st == 0
            if mode == "binary":
                og = ro.home_objective(w.cost, oh, r["S"], pe_old, ps, gm, w.kappa)
                orf = ro.home_objective(w.cost, oh, p, pe_old, ps, gm, w.kappa)
                err = np.max(np.abs(og - orf)[ok] / np.maximum(1, np.abs(orf[ok])))
                same = (np.abs(r["S"] - p).max(1) == 0)[ok].mean()
                worst[(mode, "obj")] = max(worst.get((mode, "obj"), 0), err)
                worst[(mode, "1-same")] = max(worst.get((mode, "1-same"), 0), 1 - same)
            else:
                err = np.abs(r["S"] - p)[ok].max()
                worst[(mode, "p")] = max(worst.get((mode, "p"), 0), err)
            chk = pe_new - (r["S"] + w.load)
            derr = np.abs(r["diff"] - np.linalg.norm(chk, axis=1) / T).max()
            worst[(mode, "diff")] = max(worst.get((mode, "diff"), 0), derr)
print({k: float(f"{v:.3g}") for k, v in worst.items()})
