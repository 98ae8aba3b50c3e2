"""The dense f64 product R p of the operator (gemm_tn_kernel on the matrix cores), 200 launches -- the
path of feeders given only as a matrix and of Newton evaluations on feeders beyond the tree form
(for rocprofv3 --pmc: MFMA counters of the matvec).  python tools/matvec_run.py [--T 24] [--nodes 2048]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine, OperatorOptions  # noqa: E402
from revs_admm_amd.synthetic import make_workload            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=24)
ap.add_argument("--nodes", type=int, default=2048)
a = ap.parse_args()
w = make_workload(20000, a.T, n_nodes=a.nodes, seed=0, binary_feasible=False)
e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh,
               mode="pdhg", op=OperatorOptions(voltage="dense"))
e.pnq.uniform_(0.0, 3.0)
for _ in range(200):
    e._gemm1(e.R64T, e.pnq[0], e.v_sl)
torch.cuda.synchronize()
print("done")
