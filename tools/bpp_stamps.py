"""Stage stamps of op_dual_bpp_kernel's last launch on the 121144 feeder (tuning build:
python -m revs_admm_amd.build --out tune/librevs_bpp.so -DREVS_TUNING -DREVS_BPP_STAMPS; REVS_LIB=tune/librevs_bpp.so python tools/bpp_stamps.py)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import golden_homes
from oracle import revs_oracle as ro
from revs_admm_amd import _lib
from revs_admm_amd.engine import AdmmEngine, OperatorOptions, pack_homes
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
assert lib.revs_tuning_bpp_stamps(buf) == 0
h = np.frombuffer(buf, dtype=np.float64).reshape(256, 32)[:24]
rel = (h[:, :27] - h[:, :1]) * 0.01
print("shader clock inside the launch: %.0f MHz" % ((h[0, 27] - h[0, 28]) / ((h[0, 31] - h[0, 0]) * 0.01)))
worst = int(np.argmax(h[:, 31] - h[:, 0]))
print("slowest slot", worst, "rows", h[worst, 30], "rounds", h[worst, 29], "total us", (h[worst, 31] - h[worst, 0]) * 0.01)
print("stamps us (1 slabs summed | 2 set up | per round: lists, factored, solved, judged):", np.round(rel[worst, 1:3 + 4 * int(h[worst, 29])], 1))
print("round 0's solves: start %.1f, forward done %.1f, backward done %.1f us" % tuple(rel[worst, 20:23]))
