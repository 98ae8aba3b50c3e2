#!/usr/bin/env python3
"""Benchmark of the REVS ADMM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--homes H] [--T 24] [--mode pdhg]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(one rank per GPU, RCCL).  A "step" is ONE ADMM iteration of lpsolver.solve_ADMM
(reference lpsolver.py:254-287) over all residences: the operator QP, every home
QP (batched PDHG kernel), the dual update and the residual reduction.  Weak
scaling: every GPU owns `--homes` residences (default 100 000 x T=24, the size
BASELINE.json's metric is quoted on).  The timed steps start `--spinup` (30) iterations
into the ADMM run: the first ~10 iterations are a transient in which voltage rows bind
hard and residences are clamped (operator QP: a few Newton iterations on its dual);
afterwards R.(aggregate load) respects every row and the operator side is one evaluation
of its dual -- home pass, one f64 product, row check.  The transient's cost is reported
beside the headline.  The feeder's 2048 constraint nodes are replicated and the only
collective is the all-reduce of the node aggregate (once per evaluation: once per ADMM
iteration in the steady state).

After the W warm-up steps a burst of untimed throw-away products (`--clock-warm`, ~25 ms,
no ADMM state touched) brings the GPU to steady clocks; the K timed steps are
`AdmmEngine.run_steps(K)` = K x `step()`, consecutive steady-state iterations inside one
native call.

Prints ONE JSON line (rank 0).  `value` = home-QP solves per second, whole job,
inputs resident in HBM, from K steps with nothing else on the stream; the per-kernel
durations come from an instrumented repeat of the same K steps (HIP event records cost GPU
time).  `roofline` is the PDHG sweep kernel against HBM;
`roofline_matvec` the f64 matrix-core product of the operator against the f64 MFMA
peak; `cpu_baseline` the oracle (numpy port of the reference algorithm) timed on
this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)
F64_MFMA_PEAK_TFLOPS = 78.6    # MI355X datasheet, dense f64 matrix


def agent_bytes_per_home(T, write_sc, pdhg_dual, fused=False, recompute=False):
    """Algorithmic HBM bytes of revs_agent_step per residence:
    reads LOAD, P_est[k], P_est[k+1], P_sch[k], G[k] (5 profiles) + the 32-byte home
    record; writes P_sch[k+1], G[k+1] (2 profiles) + diff + dsq + status (12 bytes);
    the PDHG multipliers when they are carried across iterations (one float per home read
    and written; one profile each way with full_rows); S and C (2T+1 floats) only on the
    iteration whose schedules are returned; with the next operator home pass folded in
    (`fused`) the node index (4 bytes) and the P_est[k+2] candidate (one profile written)."""
    b = 5 * 4 * T + 32 + 2 * 4 * T + 12
    if pdhg_dual == "full":
        b += 2 * 4 * T
    elif pdhg_dual:
        b += 2 * 4
    if write_sc:
        b += 4 * T + 4 * (T + 1)
    if fused:       # the sweep also does the next operator home pass: node index in, P_est out
        b += 4 + 4 * T
    if recompute:   # ... and recomputes P_est[k+1] instead of reading it
        b -= 4 * T
    return b


def cpu_baseline(w, budget_s=20.0):
    """Oracle (numpy restatement of lpsolver.py) on the host: full ADMM iterations --
    operator QP + home QPs + dual update -- on the first `ns` residences of the same
    workload and the nodes they touch.  One process; numpy's BLAS may use several
    threads for the operator's dense algebra, the home solves are single-threaded."""
    from oracle import revs_oracle as ro
    ns = min(w.N, 4096)
    ms = int(w.node_of[ns - 1]) + 1
    import copy
    ws = copy.copy(w)
    ws.load, ws.homes, ws.node_of, ws.Rn = w.load[:ns], w.homes[:ns], w.node_of[:ns], w.Rn[:ms, :ms]
    oh = ro.homes_from_records(ws.load, ws.homes)
    iters = 0
    t0 = time.perf_counter()
    # solve_ADMM keeps its own state; time successive whole runs of 2 iterations
    while True:
        ro.solve_ADMM(oh, ws.Rn, ws.node_of, ws.cost, ws.kappa, 2, ws.vset, ws.vlow, ws.vhigh,
                      mode="relaxed", util_eps=1e-8)
        iters += 2
        dt = time.perf_counter() - t0
        if dt > budget_s or iters >= 6:
            break
    try:
        import threadpoolctl
        cores = max([p["num_threads"] for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        cores = 1
    return {"value": ns * iters / dt, "unit": "solves/s", "cores": int(cores), "kind": "port",
            "sample": f"{iters} ADMM iterations (operator QP + home QPs + dual update) on the "
                      f"first {ns} residences / {ms} nodes of the same synthetic workload, "
                      f"oracle/revs_oracle.py (numpy float64), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup", type=int, default=30,
                    help="ADMM iterations run (untimed, but reported) before the warm-up so that "
                         "the timed steps start from a mid-run state")
    ap.add_argument("--homes", type=int, default=100_000, help="residences per GPU")
    ap.add_argument("--T", type=int, default=24)
    ap.add_argument("--nodes", type=int, default=2048)
    ap.add_argument("--mode", default="pdhg", choices=["pdhg", "relaxed_exact", "binary"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--stress", type=float, default=1.0,
                    help="coordinated-profile voltage / limit of the synthetic feeder")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clock-warm", type=int, default=2000,
                    help="untimed throw-away products enqueued after the warm-up steps so that the "
                         "timed region starts at steady GPU clocks (0: none)")
    ap.add_argument("--eps", type=float, default=1e-4, help="ADMM residual target")
    ap.add_argument("--op-check", type=int, default=None, help="operator: residual test period")
    ap.add_argument("--op-eps", type=float, default=None, help="operator: stopping tolerance")
    ap.add_argument("--op-rho-v", type=float, default=None, help="operator: rho_v scale")
    ap.add_argument("--op-rho-b", type=float, default=None, help="operator: rho_b scale")
    ap.add_argument("--op-alpha", type=float, default=None, help="operator: over-relaxation")
    ap.add_argument("--op-adapt", type=int, default=None, help="operator: rho update period")
    ap.add_argument("--pdhg-check", type=int, default=None, help="PDHG: convergence test period")
    ap.add_argument("--op-kadd", type=int, default=None,
                    help="operator: violated rows admitted to a slot's model per Newton iteration")
    ap.add_argument("--no-converge", action="store_true",
                    help="skip the untimed run to the eps-residual (profiling runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    group = None
    if world > 1 or os.environ.get("REVS_FORCE_GROUP"):      # (1-rank group: rehearsal of N>1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device(dev))
        group = dist.group.WORLD

    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload

    n_total = args.homes * world
    w = make_workload(n_total, args.T, n_nodes=args.nodes, seed=args.seed,
                      binary_feasible=(args.mode == "binary"), stress=args.stress)
    lo, hi = w.shard(rank, world)
    counts = np.bincount(w.node_of, minlength=w.M)
    opts = OperatorOptions()
    if args.op_check:
        opts.check_every = args.op_check
        opts.adapt_every = max(opts.check_every, (100 // opts.check_every) * opts.check_every)
    if args.op_eps:
        opts.eps = args.op_eps
    if args.op_rho_v:
        opts.rho_v_scale = args.op_rho_v
    if args.op_rho_b:
        opts.rho_b_scale = args.op_rho_b
    if args.op_alpha:
        opts.alpha = args.op_alpha
    if args.op_adapt is not None:
        opts.adapt_every = args.op_adapt
    if args.op_kadd is not None:
        opts.newton_kadd = args.op_kadd
    if os.environ.get("REVS_CAL_RHO_B"):
        opts.cal_rho_b = tuple(float(x) for x in os.environ["REVS_CAL_RHO_B"].split(","))
    if os.environ.get("REVS_CAL_RHO_V"):
        opts.cal_rho_v = tuple(float(x) for x in os.environ["REVS_CAL_RHO_V"].split(","))
    eng = AdmmEngine(w.cost, w.homes[lo:hi], w.load[lo:hi], w.node_of[lo:hi], w.Rn, kappa=w.kappa,
                     vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode=args.mode, device=dev,
                     group=group, node_counts=counts, op=opts,
                     pdhg=({"check": args.pdhg_check} if args.pdhg_check else None))
    n_local = hi - lo

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()       # ranks leave the set-up together (a step waits for every rank's all-reduce)

    # Spin-up: the first ADMM iterations are a transient of their own -- every charger
    # jumps to the cheapest slots, voltage rows bind hard and some residences are clamped,
    # so the operator QP needs hundreds to thousands of inner iterations.  They are run
    # here, outside the timed region, and reported under breakdown.transient.
    spin_ms = []
    for _ in range(args.spinup):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.step(write_sc=False)
        torch.cuda.synchronize()
        spin_ms.append((time.perf_counter() - t1) * 1e3)
    spin_inner = list(eng.op_iters_hist)
    spin_paths = "".join(p[0] for p in eng.op_path_hist)
    for _ in range(args.warmup):
        eng.step(write_sc=False)
    # The spin-up above synchronises after every iteration and W steps are a fraction of a
    # millisecond: the GPU would enter the timed region (7 ms at the defaults) at idle clocks
    # (measured: 0.0370 vs 0.0344 ms per step on the same box).  Keep it busy for ~25 ms with
    # untimed throw-away work -- voltage products on scratch operands, no ADMM state touched.
    for _ in range(args.clock_warm):
        eng._gemm1(eng.R64T, eng.pnq[2], eng.v_sl)
    barrier()
    inner0 = len(eng.op_iters_hist)
    spec0 = list(eng.spec_hist)
    # ---- the timed region: exactly K steps, nothing else on the stream ----
    t0 = time.perf_counter()
    eng.run_steps(args.steps)       # = args.steps x eng.step(write_sc=False), see engine.py
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    # ---- the same K steps again, instrumented: two HIP events per step on the stream the
    # kernels run on (between the operator part and the home sweep, and after the sweep; the
    # previous step's last event opens the operator part).  An event record costs ~3 us of
    # GPU time on this stack, so the per-kernel durations come from this repeat and `value`
    # from the plain loop above; the instrumented loop's own ms_per_step is reported too.
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps + 1)]
    barrier()
    evs[0][2].record()
    t1 = time.perf_counter()
    for k in range(args.steps):
        evs[k + 1][0] = evs[k][2]
        eng.step(write_sc=False, events=evs[k + 1])
    barrier()
    dt_instr = time.perf_counter() - t1
    evs = evs[1:]
    rp, rd, dmax, conv = eng.residuals(args.eps)
    agent_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
    oper_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    inner = eng.op_iters_hist[inner0:]
    # The sweep's own launch duration without the records' overhead inside the interval: the
    # steady-state launch (selection workgroups and folded home pass included) replayed
    # back to back on the current state, two events around the whole batch.
    agent_b2b_ms = None
    if getattr(eng, "_plan", None) is not None and eng._fused_ready:
        fused = True
        p_scr = torch.zeros_like(eng.pnq[0])
        pe_scr = torch.zeros_like(eng.P_est)
        nrep = 100
        # (every replay starts from its own copy of the homes' warm-start multipliers, so
        # that each one does the work of the real step)
        duals = [None] * (nrep + 5) if eng.pdhg_dual is None else \
            [eng.pdhg_dual.clone() for _ in range(nrep + 5)]
        for i in range(5):
            eng.replay_sweep(p_scr, pe_scr, fused, duals[nrep + i])
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        for i in range(nrep):
            eng.replay_sweep(p_scr, pe_scr, fused, duals[i])
        r1.record()
        torch.cuda.synchronize()
        agent_b2b_ms = r0.elapsed_time(r1) / nrep
        del duals
    st = eng.status.cpu().numpy()
    pdhg_it = float((st >> 8)[(st >> 8) > 0].mean()) if args.mode == "pdhg" and ((st >> 8) > 0).any() else None

    # the f64 matrix-core product of the operator path that actually ran: the voltage rows
    # R.p of the dual Newton path (M x M x T), Rs.p0 / Q^T w (M x M x T) on the node-space
    # ADMM fast path, Q^T [rhat | w] (2T columns) on the general ADMM path
    path = eng.op_path_hist[-1]
    fast = path in ("node", "dual")
    reps = 200
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        if path == "dual":
            eng._gemm1(eng.R64T, eng.pnq[0], eng.v_sl)
        elif fast:
            eng._gemm1(eng.Rs, eng.p0, eng.f_wh)
        else:
            eng._gemm_cat(eng.Q, eng.rhat, eng.w, eng.ta, eng.tb)
    e1.record()
    torch.cuda.synchronize()
    gemm_ms = e0.elapsed_time(e1) / reps
    gemm_flops = 2.0 * eng.M * eng.M * args.T * (1 if fast else 2)

    # how many ADMM iterations until the eps-residual (continues the same run)
    iters_to_eps = None
    if args.mode != "binary" and not args.no_converge:
        k = eng.iteration
        while k < 600:
            rp, rd, dmax, conv = eng.residuals(args.eps)
            if conv:
                iters_to_eps = k
                break
            eng.step(write_sc=False)
            k += 1

    if rank == 0:
        warm = (None if eng.pdhg_dual is None else
                ("full" if eng.pdhg_dual.dim() == 2 else "scalar"))
        bph = agent_bytes_per_home(args.T, False, warm, fused=bool(getattr(eng, "_fused_ready", False)),
                                   recompute=bool(getattr(eng, "recompute_pe_new", False)))
        bytes_per_launch = bph * n_local
        launch_ms = agent_b2b_ms if agent_b2b_ms is not None else agent_ms
        ach = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "agent_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if (tj.get("homes") == n_local and tj.get("T") == args.T
                        and tj.get("mode") == args.mode
                        and tj.get("algorithmic_bytes_per_launch") == bytes_per_launch):
                    traffic = tj["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "agent_qp_solves_per_sec",
            "value": n_total * args.steps / dt,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_instrumented": dt_instr / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic {args.homes} homes/GPU x T={args.T} box+SOC home QP, "
                            f"{args.nodes}-node radial feeder, one ADMM iteration per step "
                            "(operator QP by dual Newton + all home QPs + dual update + "
                            f"residuals), timed from a state {args.spinup + args.warmup} ADMM "
                            "iterations into the run; the transient before it is in "
                            "breakdown.transient",
                "homes_per_gpu": args.homes, "homes_total": n_total, "T": args.T,
                "nodes": args.nodes, "home_solver": args.mode, "kappa": w.kappa,
                "operator_dtype": "f64", "parallelism": f"homes sharded x{world}, nodes replicated",
                "clock_warmup_products": args.clock_warm,
            },
            "roofline": {
                "kernel": "agent_step_kernel (home QP sweep + dual update"
                          + (" + next operator home pass" if getattr(eng, "_fused_ready", False) else "") + ")",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "bytes_per_home": bph,
                # back to back = the kernel (rocprofv3 agrees); the in-loop figure is one
                # event pair around one launch and includes ~3 us of record overhead
                "avg_launch_ms": launch_ms, "avg_launch_ms_in_loop_event_pair": agent_ms,
                "pdhg_iters_mean": pdhg_it,
            },
            "roofline_matvec": {
                "kernel": "gemm_tn_kernel<double> (" + ("voltage rows R.p, M x M x T" if path == "dual" else "voltage check Rs.p0, M x M x T" if fast else "Q^T [rhat | w], M x M x 2T") + ")",
                "bound": "mfma", "achieved": gemm_flops / (gemm_ms * 1e-3) / 1e12,
                "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": gemm_flops / (gemm_ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                "avg_launch_ms": gemm_ms,
                # 6 flop per byte of R at T = 24: the product streams the matrix, so its own
                # roofline is HBM -- reported beside the matrix-core utilisation
                "matrix_stream_GBs": 8.0 * eng.M * eng.M / (gemm_ms * 1e-3) / 1e9,
                "matrix_stream_frac_of_hbm_peak": 8.0 * eng.M * eng.M / (gemm_ms * 1e-3) / 1e9
                                                  / HBM_PEAK_GBS,
            },
            "breakdown": {
                "operator_ms_per_step": oper_ms, "agent_ms_per_step": agent_ms,
                "operator_inner_iters_mean": float(np.mean(inner)) if inner else 0.0,
                "operator_path": eng.op_path_hist[-1],
                "operator_newton_iters_mean": (float(np.mean([h[0] for h in eng.newton_hist[-args.steps:]]))
                                               if eng.newton_hist else None),
                "speculative_sweeps_kept_discarded": [eng.spec_hist[0] - spec0[0],
                                                      eng.spec_hist[1] - spec0[1]],
                "operator_voltage_rows": int(eng.M * args.T),
                "admm_residual_primal": rp, "admm_residual_dual": rd, "admm_max_diff": dmax,
                "admm_iters_to_eps": iters_to_eps, "eps": args.eps,
                # whole-run view: transient + steady iterations of a 400-iteration run
                "amortized_400_iterations_solves_per_sec":
                    (n_total * 400 / ((np.sum(spin_ms) + (400 - args.spinup) * dt / args.steps * 1e3)
                                      * 1e-3)) if spin_ms and args.spinup <= 400 else None,
                "transient": {
                    "spinup_iterations": args.spinup,
                    "ms_per_step_mean": float(np.mean(spin_ms)) if spin_ms else None,
                    "ms_per_step_max": float(np.max(spin_ms)) if spin_ms else None,
                    "ms_total": float(np.sum(spin_ms)) if spin_ms else None,
                    "operator_inner_iters": spin_inner,
                    "operator_paths": spin_paths,
                },
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if group is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
