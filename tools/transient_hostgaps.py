"""Host time between the native calls of the transient (AdmmEngine.run_steps from the zero state, third engine of the
process): every library entry point the driver uses is wrapped with two time stamps."""
import os
import sys
import time

for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine          # noqa: E402
from revs_admm_amd.synthetic import make_workload    # noqa: E402

w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12      # 0: AdmmEngine.run(1000, eps=1e-4, history=False) instead
mode = sys.argv[2] if len(sys.argv) > 2 else "pdhg"
log = []


class Proxy:
    def __init__(self, lib):
        self._lib = lib
        self._w = {}

    def __getattr__(self, k):
        f = getattr(self._lib, k)
        if not k.startswith("revs_") or not callable(f):
            return f
        if k not in self._w:
            def wrapped(*a, _f=f, _k=k):
                t0 = time.perf_counter()
                r = _f(*a)
                log.append((_k, t0, time.perf_counter()))
                return r
            self._w[k] = wrapped
        return self._w[k]


for rep in range(3):
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh,
                   mode=mode, feeder=w.feeder)
    if rep == 2:
        e.lib = Proxy(e.lib)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = e.run(1000, eps=1e-4, history=False) if n == 0 else (e.run(n, history=False) if mode == "binary" else e.run_steps(n))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("engine", rep, "%s iterations: host %.3f ms, with the closing synchronize %.3f ms" % (its if n == 0 else n, (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3), flush=True)
prev = t0
tot_py = tot_nat = 0.0
for k, a, b in log:
    print("  python %6.1f us | %-34s %7.1f us" % ((a - prev) * 1e6, k, (b - a) * 1e6))
    tot_py += a - prev
    tot_nat += b - a
    prev = b
print("  python %6.1f us (tail)" % ((t1 - prev) * 1e6))
print("python between native calls %.1f us, inside native calls %.1f us" % ((tot_py + t1 - prev) * 1e6, tot_nat * 1e6))
