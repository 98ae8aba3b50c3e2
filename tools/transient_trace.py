"""Kernel timeline of the transient as ONE call (AdmmEngine.run_steps(12) from the zero state, second engine of the
process): run under rocprofv3 --kernel-trace, then `python tools/transient_trace.py --read <dir>` prints busy time,
gaps and the launches in order.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -o tt -- python3 tools/transient_trace.py
    python tools/transient_trace.py --read gpurun_out/tt"""
import glob
import os
import sys
import time

if "--read" in sys.argv:
    import csv
    d = sys.argv[sys.argv.index("--read") + 1]
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    # the marked window: between the two launches of the marker product (a gemm on scratch operands, 7 in a row)
    names = [r[2] for r in rows]
    marks = [i for i in range(len(rows) - 6) if all("gemm_tn_kernel<double" in names[i + k] for k in range(7))]
    a, b = marks[-2] + 7, marks[-1]
    win = rows[a:b]
    t0, t1 = win[0][0], win[-1][1]
    busy = sum(e - s for s, e, _ in win)
    print("launches %d, window %.1f us, kernels busy %.1f us (%.0f %%)" % (len(win), (t1 - t0) / 1e3, busy / 1e3, 100.0 * busy / (t1 - t0)))
    agg = {}
    for s, e, n in win:
        k = n.split("(")[0][:70]
        agg.setdefault(k, [0, 0.0])
        agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
    for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-70s %4d  %8.1f us" % (k, c, us))
    gaps = [(win[i + 1][0] - win[i][1]) / 1e3 for i in range(len(win) - 1)]
    print("gaps between launches: sum %.1f us, > 5 us: %s" % (sum(g for g in gaps if g > 0), [round(g, 1) for g in gaps if g > 5]))
    if "--list" in sys.argv:
        for (s, e, n), g in zip(win, [0.0] + gaps):
            print("  +%7.1f us  gap %5.1f  %6.1f us  %s" % ((s - t0) / 1e3, g, (e - s) / 1e3, n.split("(")[0][:80]))
    sys.exit(0)

for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import AdmmEngine          # noqa: E402
from revs_admm_amd.synthetic import make_workload    # noqa: E402

w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for rep in range(2):
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh,
                   mode="pdhg", feeder=w.feeder)
    for _ in range(200):
        e._gemm1(e.R64T, e.pnq[2], e.v_sl)
    torch.cuda.synchronize()
    for _ in range(7):
        e._gemm1(e.R64T, e.pnq[2], e.v_sl)           # (marker)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.run_steps(n)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    for _ in range(7):
        e._gemm1(e.R64T, e.pnq[2], e.v_sl)           # (marker)
    torch.cuda.synchronize()
    print("engine", rep, "%d iterations as one call %.3f ms" % (n, ms), "evaluations", e.op_iters_hist[:n], flush=True)
