"""How much of the voltage product R.p can hide behind the home sweep (MI355X)?  Times, from
a steady-state ADMM state, K launches of the f64 product alone, of the sweep alone, of both
alternating on one stream, and of both on two streams with no dependency between them.
python tools/overlap_probe.py [homes] [T]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine, ptr, check   # noqa: E402
from revs_admm_amd.synthetic import make_workload         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
K = 300
w = make_workload(n, T, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
               vlow=w.vlow, vhigh=w.vhigh, mode="pdhg")
for _ in range(35):
    e.step(write_sc=False)
torch.cuda.synchronize()
s_main = e.stream
side = torch.cuda.Stream()


def gemm(st):
    check(e.lib.revs_gemm_tn_f64_split(e.M, e.T, e.M, ptr(e.R64T), ptr(e.pnq[0]), ptr(e.v_sl),
                                       e.ksplit1, st), "gemm")


def sweep():
    e.agent_step(write_sc=False, to_alt=True)


def timed(label, fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print(f"{label:34s} {(time.perf_counter() - t0) / K * 1e6:7.2f} us per iteration", flush=True)


timed("product alone", lambda: [gemm(s_main) for _ in range(K)])
timed("sweep alone", lambda: [sweep() for _ in range(K)])
timed("product, sweep on one stream", lambda: [(gemm(s_main), sweep()) for _ in range(K)])
timed("product | sweep on two streams", lambda: [(gemm(side.cuda_stream), sweep()) for _ in range(K)])
