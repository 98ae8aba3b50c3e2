"""Stage stamps of the last stream_block_verdict launch of a burst of 20 iterations (tuning build:
python -m revs_admm_amd.build --out tune/librevs_vd.so -DREVS_TUNING -DREVS_VD_STAMPS;
REVS_LIB=tune/librevs_vd.so python tools/verdict_stamps.py [steps])."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from revs_admm_amd import _lib
from revs_admm_amd.engine import AdmmEngine
from revs_admm_amd.synthetic import make_workload
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh,
               mode="pdhg", feeder=w.feeder)
e.run_steps(40)
for _ in range(5):
    e.run_steps(steps)
    torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_double * (1024 * 8))()
assert lib.revs_tuning_verdict_stamps(buf) == 0
h = np.frombuffer(buf, dtype=np.float64).reshape(1024, 8)
# (the last launch only: its workgroups start within a microsecond of each other; rows of earlier, larger launches stay behind)
live = h[:, 0] > h[:, 0].max() - 500.0
n = int(live.sum())
t0 = h[live, 0].min()
rel = (h[live, :8] - t0) * 0.01
print("workgroups stamped", n)
print("start: min %.1f median %.1f max %.1f us" % (rel[:, 0].min(), np.median(rel[:, 0]), rel[:, 0].max()))
for i, name in ((1, "control word read"), (2, "rows judged / slice copied"), (3, "counted")):
    d = rel[:, i] - rel[:, 0]
    print("%-28s since the workgroup's start: median %.1f max %.1f us; since the launch's first start: max %.1f" % (name, np.median(d), d.max(), rel[:, i].max()))
jd = rel[:n - 32]
print("two-slot workgroups: first scan done median %.1f, second scan done median %.1f us since the workgroup's start" % (
    np.median(jd[:, 6] - jd[:, 0]), np.median(jd[:, 7] - jd[:, 0])))
last = int(np.argmax(np.where(h[live, 5] > h[live, 0], h[live, 5], 0.0)))
print("last arriver: records from %.1f to %.1f us" % (rel[last, 4], rel[last, 5]))
ex = rel[n - 32:n]                     # (kHandOverGroups workgroups at the end of the grid)
if len(ex):
    print("hand-over workgroups (%d): copied at %.1f .. %.1f us since the launch's first start (own duration median %.1f)" % (
        len(ex), ex[:, 2].min(), ex[:, 2].max(), np.median(ex[:, 2] - ex[:, 0])))
