"""Where the time of a short burst goes (the driver benches with --steps 20: 0.37 ms of kernels):
host time before, inside and after the native call, the closing synchronize, and the kernels' own
time by the library's events.  python tools/burst_cost.py"""
import os, sys, time
for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine, OperatorOptions
from revs_admm_amd.synthetic import make_workload
import revs_admm_amd.steady_state as ss
w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode="pdhg", feeder=w.feeder)
for _ in range(30):
    e.step(write_sc=False)
e.run_steps(5)
marks = {}
lib = e.lib
orig = lib.revs_plan_stream_run_blocks
def native(*a):
    marks["n0"] = time.perf_counter()
    r = orig(*a)
    marks["n1"] = time.perf_counter()
    return r
class L:      # proxy so that only this entry point is wrapped
    def __getattr__(self, k):
        return native if k == "revs_plan_stream_run_blocks" else getattr(lib, k)
e.lib = L()
res = []
for rep in range(12):
    torch.cuda.synchronize()
    for _ in range(200):       # keep clocks up as bench's clock_warm does
        e._gemm1(e.R64T, e.pnq[2], e.v_sl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes as C
    lib.revs_plan_stream_timing(e._plan, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ta = time.perf_counter()
    e.run_steps(20)
    tb = time.perf_counter()
    tc = time.perf_counter()
    torch.cuda.synchronize()
    td = time.perf_counter()
    ms = C.c_double()
    lib.revs_plan_stream_elapsed_ms(e._plan, C.addressof(ms))
    res.append([(ta - t0), (marks["n0"] - ta), (marks["n1"] - marks["n0"]), (tb - marks["n1"]), (tc - tb), (td - tc), (td - t0), ms.value * 1e-3])
r = np.array(res) * 1e6
print("us: e0.record | python before native | native call | python after | e1.record | synchronize | total | events")
for row in r:
    print(" ".join(f"{x:8.1f}" for x in row))
print("median", " ".join(f"{x:8.1f}" for x in np.median(r, axis=0)), " -> per step", round(float(np.median(r[:, 6])) / 20, 2))
