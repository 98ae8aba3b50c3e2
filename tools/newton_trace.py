"""Per-evaluation trace of the operator's Newton solves (Python loop): violated rows without a multiplier, rows with one, rows
admitted -- on the 121144 feeder's 15 iterations (on/off chargers) and on the bench workload's transient, for several
admission rules.  And the decay of max_h diff along a run to the eps-residual (what AdmmEngine.run(eps) sizes its bursts by).
    python tools/newton_trace.py"""
import os, sys
for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from revs_admm_amd.engine import AdmmEngine, OperatorOptions, pack_homes
from revs_admm_amd.synthetic import make_workload
from oracle import revs_oracle as ro
from conftest import golden_homes

def show(label, e):
    print(label)
    last = None
    for it, nw, nvs, nvm, nss, nsm, kadd, rmax in e.newton_trace:
        if it != last:
            print(f"  ADMM iteration {it + 1}:")
            last = it
        print(f"    newton {nw:2d}: violated sum {nvs:5d} max {nvm:4d} | multipliers sum {nss:5d} max {nsm:3d} | kadd {kadd:2d} | rmax {rmax:.2e}")

z, fd = ro.load_golden(os.path.join(ROOT, "tests", "golden", "revs_121144.npz"))
R = ro.compute_Rmat_tree(fd); nonsub, res = fd.nonsub(), fd.res()
pos = -np.ones(fd.n_nodes, np.int64); pos[nonsub] = np.arange(len(nonsub)); Rr = R[np.ix_(pos[res], pos[res])]
oh, evi = golden_homes(z, "dis_a90_r4800", 4.8); n = oh.LOAD.shape[0]
for cold in (0, 16):
    e = AdmmEngine(z["tariff_shift6"], pack_homes(oh.ev, 4.8, 20.0, 0.2, 11, 23), oh.LOAD, np.arange(n), Rr, kappa=5.0, vset=1.03,
                   vlow=0.95, vhigh=1.05, mode="binary", op=OperatorOptions(native_newton=False, newton_trace=True, newton_kadd_cold=cold))
    e.run(6)
    show(f"121144 feeder, on/off chargers, cold admission {cold}", e)
w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=True, stress=1.0)
for cold in (0, 16):
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode="binary",
                   feeder=w.feeder, op=OperatorOptions(native_newton=False, newton_trace=True, newton_kadd_cold=cold))
    e.run(8)
    show(f"synthetic 100 000 x 24, on/off chargers, cold admission {cold}", e)
w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode="pdhg", feeder=w.feeder)
e.run(700, history=False)
md = e.max_diff
print("max diff along the PDHG run (iteration: value), every 10th:", {k: float(f"{md[k]:.3e}") for k in sorted(md) if k % 10 == 0})
print("stream calls", e.stream_calls)
