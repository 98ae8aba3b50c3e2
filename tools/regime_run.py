"""One regime of bench.py on its own (for rocprofv3 --kernel-trace --stats):

    python tools/regime_run.py --regime binding|binary|steady [--homes 100000] [--T 24] [--steps 300]

binding: PDHG residences, stress 1.3 (rows keep binding: one chained Newton iteration per ADMM
iteration); binary: the reference's binary chargers at stress 1.0; steady: the headline regime."""
import argparse
import os
import sys
import time

for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine, OperatorOptions  # noqa: E402
from revs_admm_amd.synthetic import make_workload            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--regime", default="binding")
ap.add_argument("--homes", type=int, default=100_000)
ap.add_argument("--T", type=int, default=24)
ap.add_argument("--nodes", type=int, default=2048)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--spin", type=int, default=60)
ap.add_argument("--kadd", type=int, default=None, help="violated rows admitted to a slot's model per Newton iteration")
ap.add_argument("--voltage", default="auto", help="auto (tree form where the feeder is known) | dense (the f64 product on the matrix cores)")
a = ap.parse_args()
mode, stress = {"binding": ("pdhg", 1.3), "binary": ("binary", 1.0), "steady": ("pdhg", 1.0)}[a.regime]
w = make_workload(a.homes, a.T, n_nodes=a.nodes, seed=0, binary_feasible=(mode == "binary"), stress=stress)
e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow,
               vhigh=w.vhigh, mode=mode, feeder=w.feeder,
               op=OperatorOptions(voltage=a.voltage, **({"newton_kadd": a.kadd} if a.kadd is not None else {})))
e.run_steps(a.spin)
torch.cuda.synchronize()
c0, s0 = list(e.chain_hist), list(e.spec_hist)
t0 = time.perf_counter()
e.run_steps(a.steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{a.regime}: {dt / a.steps * 1e3:.4f} ms per iteration over {a.steps}; chained kept/redone "
      f"{e.chain_hist[0] - c0[0]}/{e.chain_hist[1] - c0[1]}, streamed kept/discarded {e.spec_hist[0] - s0[0]}/"
      f"{e.spec_hist[1] - s0[1]}, evaluations per step {sum(e.op_iters_hist[-a.steps:]) / a.steps:.2f}, "
      f"Newton steps taken inside the folded chain beyond the first {e.fold_steps}")
