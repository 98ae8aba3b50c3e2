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
ap.add_argument("--config3", action="store_true",
                help="BASELINE config 3's own shape: the 121144 feeder's 1 126 residence rows, T = 96 (tests/golden/revs_121144.npz)")
a = ap.parse_args()
if a.config3:
    import networkx as nx
    import numpy as np
    from revs_admm_amd.engine import pack_homes
    from revs_admm_amd.lpsolver import compute_Rmat
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(ROOT, "tests", "golden", "revs_121144.npz"))
    g = nx.Graph()
    for nid, lab in zip(z["node_id"], z["node_label"]):
        g.add_node(int(nid), label=lab.decode())
    for u, v, r in zip(z["edge_u"], z["edge_v"], z["edge_r"]):
        g.add_edge(int(z["node_id"][u]), int(z["node_id"][v]), r=float(r))
    res = [n for n in g if g.nodes[n]["label"] == "H"]
    pos = {n: i for i, n in enumerate(n for n in g.nodes if g.nodes[n]["label"] != "S")}
    ri = [pos[n] for n in res]
    R_res = compute_Rmat(g)[np.ix_(ri, ri)]
    row = {int(h): i for i, h in enumerate(z["res_id"])}
    load = np.repeat(np.stack([z["LOAD"][row[h]] for h in res]), 4, axis=1)
    e = AdmmEngine(np.repeat(z["tariff_shift6"], 4), pack_homes(np.ones(len(res), bool), 4.8, 20.0, 0.2, 44, 92), load,
                   np.arange(len(res)), R_res, kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode="binary",
                   op=OperatorOptions(voltage="dense"))
    print("config 3 shape: M =", e.M, "T =", e.T)
else:
    w = make_workload(20000, a.T, n_nodes=a.nodes, seed=0, binary_feasible=False)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh,
                   mode="pdhg", op=OperatorOptions(voltage="dense"))
e.pnq.uniform_(0.0, 3.0)
for _ in range(200):
    e._gemm1(e.R64T, e.pnq[0], e.v_sl)
torch.cuda.synchronize()
print("done")
