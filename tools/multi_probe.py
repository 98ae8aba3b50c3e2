"""Multi-iteration sweep (revs_agent_step_multi) against one launch per iteration: same bits, and
the launch time per ADMM iteration for kin = 1 .. REVS_AGENT_MAX_INNER.

    python tools/multi_probe.py [--homes 100000] [--T 24] [--mode pdhg] [--reps 50]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd._lib import check, ptr            # noqa: E402
from revs_admm_amd.engine import AdmmEngine          # noqa: E402
from revs_admm_amd.synthetic import make_workload    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--homes", type=int, default=100_000)
    ap.add_argument("--T", type=int, default=24)
    ap.add_argument("--nodes", type=int, default=2048)
    ap.add_argument("--mode", default="pdhg")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--spin", type=int, default=40)
    a = ap.parse_args()
    w = make_workload(a.homes, a.T, n_nodes=a.nodes, seed=0, binary_feasible=(a.mode == "binary"), stress=1.0)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow,
                   vhigh=w.vhigh, mode=a.mode, feeder=w.feeder)
    e.run_steps(a.spin)
    torch.cuda.synchronize()
    lib, n, T, M = e.lib, e.n, e.T, e.M
    mt = M * T
    KM = 4
    st0 = [t.clone() for t in (e.P_est, e.P_sch, e.G)]
    y0 = e.pdhg_dual.clone() if e.pdhg_dual is not None else None

    def run(kin_list, reps=1, timed=False):
        """Apply launches with the given inner counts from the saved state; returns final state etc."""
        cur = [t.clone() for t in st0]
        nxt = [torch.empty_like(t) for t in st0]
        y = y0.clone() if y0 is not None else None
        tot = sum(kin_list)
        ring = torch.zeros(tot, mt + 64, dtype=torch.float64, device=e.dev)
        diff = torch.zeros(tot, n, dtype=torch.float32, device=e.dev)
        pen = torch.empty_like(st0[0])
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            k = 0
            for kin in kin_list:
                check(lib.revs_agent_step_multi(
                    n, T, ptr(e.cost), ptr(e.homes), ptr(e.load), ptr(cur[0]), ptr(cur[1]), ptr(cur[2]),
                    ptr(nxt[0]), ptr(nxt[1]), ptr(nxt[2]), ptr(pen), ptr(diff[k]), n, ptr(e.dsq),
                    ptr(e.status), ptr(y), ptr(y), e.kappa, e.mode, C.byref(e.pdhg), ptr(e.node_of_dev),
                    ptr(ring[k]), mt + 64, ring[k].data_ptr() + 8 * mt, kin, e.stream), "multi")
                cur, nxt = nxt, cur
                k += kin
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        return cur, y, ring, diff, pen, ms

    ref = run([1] * KM)
    for kl in ([KM], [2, 2], [3, 1], [1, 3]):
        got = run(kl)
        names = ["P_est", "P_sch", "G"]
        ok = all(torch.equal(x, y) for x, y in zip(ref[0], got[0]))
        ok_y = ref[1] is None or torch.equal(ref[1], got[1])
        ok_ring = torch.equal(ref[2][:, :mt], got[2][:, :mt])
        ok_dmax = torch.equal(ref[2][:, mt:].max(1).values, got[2][:, mt:].max(1).values)
        ok_diff = torch.equal(ref[3], got[3])
        ok_pen = torch.equal(ref[4], got[4])
        dm = got[2][:, mt:].max(1).values.cpu().numpy()
        print(f"kin={kl}: state {ok} duals {ok_y} node sums {ok_ring} dmax {ok_dmax} diff {ok_diff} next {ok_pen} "
              f"| dmax {dm} vs diff.max {got[3].max(1).values.cpu().numpy()}")
    # timing: ring contents grow with the repetitions (only the timing matters there)
    for kin in range(1, int(lib.revs_agent_max_inner(T, 0)) + 1):
        ms = run([kin], reps=a.reps, timed=True)[-1]
        print(f"kin={kin}: {ms * 1e3:.2f} us per launch, {ms * 1e3 / kin:.2f} us per ADMM iteration")


if __name__ == "__main__":
    main()
