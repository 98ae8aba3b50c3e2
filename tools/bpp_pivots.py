import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import golden_homes
from oracle import revs_oracle as ro
from revs_admm_amd.engine import AdmmEngine, OperatorOptions, pack_homes
z, fd = ro.load_golden(os.path.join(ROOT, "tests", "golden", "revs_121144.npz"))
R = ro.compute_Rmat_tree(fd)
nonsub, res = fd.nonsub(), fd.res()
pos = -np.ones(fd.n_nodes, np.int64); pos[nonsub] = np.arange(len(nonsub))
Rr = R[np.ix_(pos[res], pos[res])]
oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
n = oh.LOAD.shape[0]
e = AdmmEngine(z["tariff_shift6"], pack_homes(oh.ev, 4.8, 20.0, 0.2, 11, 23), oh.LOAD, np.arange(n), Rr, kappa=5.0, vset=1.03,
               vlow=0.95, vhigh=1.05, mode="binary", op=OperatorOptions(native_newton=False))
# instrument: record the per-slot pivot counts and candidate counts of every model call
orig = e.lib.revs_op_dual_model
log = []
import ctypes as C
def wrapped(*a):
    rc = orig(*a)
    torch.cuda.synchronize()
    log.append((np.abs(e.info_h.numpy()).copy(), e.c_cnt[0].cpu().numpy().copy(), e.c_cnt[1].cpu().numpy().copy()))
    return rc
e.lib.revs_op_dual_model = wrapped
e.run(15)
piv = np.array([l[0] for l in log])
print("model calls", len(log), "pivots per call: max over slots mean %.1f, max %d; mean over slots %.1f" % (piv.max(1).mean(), piv.max(), piv.mean()))
print("candidates max per call (either set)", [int(max(l[1].max(), l[2].max())) for l in log][:40])
print("max pivots per call", piv.max(1)[:60])
