"""Stage stamps of the evaluations' rows + selection launch (op_tree_rows_kernel<true>) on the 121144 feeder, last launch
of 8 ADMM iterations (tuning build:
python -m revs_admm_amd.build --out revs_admm_amd/tune_rows.so -DREVS_TUNING -DREVS_KV_STAMPS -DREVS_ROWS_STAMPS;
REVS_LIB=revs_admm_amd/tune_rows.so python tools/rows_stamps.py [--synthetic])."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from revs_admm_amd import _lib
from revs_admm_amd.engine import AdmmEngine, OperatorOptions, pack_homes
if "--synthetic" in sys.argv:
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh,
                   mode="pdhg", feeder=w.feeder)
    e.run_steps(6)
else:
    from conftest import golden_homes
    from oracle import revs_oracle as ro
    z, fd = ro.load_golden(os.path.join(ROOT, "tests", "golden", "revs_121144.npz"))
    R = ro.compute_Rmat_tree(fd)
    nonsub, res = fd.nonsub(), fd.res()
    pos = -np.ones(fd.n_nodes, np.int64); pos[nonsub] = np.arange(len(nonsub))
    Rr = R[np.ix_(pos[res], pos[res])]
    oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
    n = oh.LOAD.shape[0]
    e = AdmmEngine(z["tariff_shift6"], pack_homes(oh.ev, 4.8, 20.0, 0.2, 11, 23), oh.LOAD, np.arange(n), Rr, kappa=5.0, vset=1.03,
                   vlow=0.95, vhigh=1.05, mode="binary")
    e.run(8)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_double * (256 * 32))()
assert lib.revs_tuning_rows_stamps(buf) == 0
h = np.frombuffer(buf, dtype=np.float64).reshape(256, 32)[:e.T]
rel = (h - h[:, :1]) * 0.01
names = {21: "tree indices in", 22: "LDS rows cleared", 23: "(pk used)", 1: "y / q / p gathers issued, before the scans", 2: "scan 1 local", 3: "scan 1 offsets",
         4: "w' stored", 5: "scans done", 6: "rows judged", 8: "partials folded, stats out", 9: "multipliers' rows listed", 10: "arg-max rounds done", 11: "lists written (end)"}
worst = int(np.argmax(rel[:, 11]))
order = [21, 22, 23, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11]
print("slowest slot", worst, "of", e.T, "total us %.2f; mean over slots %.2f" % (rel[worst, 11], rel[:, 11].mean()))
for i in order:
    print("  %-48s slowest %6.2f   mean %6.2f" % (names[i], rel[worst, i], rel[:, i].mean()))
