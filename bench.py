#!/usr/bin/env python3
"""Benchmark of the REVS ADMM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--homes H] [--T 24] [--mode pdhg]

N > 1: one rank per GPU, the library's own RCCL communicator (bootstrapped over torch.distributed).
Either the driver starts the ranks,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
or `python bench.py --gpus N` starts them itself: with no WORLD_SIZE in the environment the parent
-- which makes no GPU call -- spawns N fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, 127.0.0.1), relays rank 0's JSON line and exits non-zero if any child does.
`--scaling weak` (default): every GPU owns `--homes` residences; `--scaling strong`: `--homes` is
the total, split over the GPUs (BASELINE config 2: 100 000 homes over 8 GPUs = 12 500 each).
`--share-gpu`: all ranks on cuda:0 with the node sums through the library's hook communicator over
gloo -- the rehearsal of the N > 1 path on a one-GPU box (RCCL refuses two ranks on one device).
A "step" is ONE ADMM iteration of lpsolver.solve_ADMM (reference lpsolver.py:254-287) over all
residences: the operator's answer and its voltage rows, every home QP (batched PDHG kernel),
the dual update and the per-home residual terms.  Weak scaling: every GPU owns `--homes`
residences (default 100 000 x T = 24, the size BASELINE.json's metric is quoted on); the
feeder's 2048 constraint nodes are replicated and the only collective is the all-reduce of
the M x T node sums -- in the steady state the sums of 32 iterations in one collective on a second
stream, beside the next 32 sweeps (verdicts by blocks).

The timed steps start `--spinup` (30) iterations into the run: the first ~10 iterations are a
transient in which voltage rows bind hard and residences are clamped (operator QP: a few
Newton iterations on its dual); afterwards the estimate respects every row and the iterations
stream (`AdmmEngine.run_steps`, revs_plan_stream_run_blocks): a sweep launch carries up to 32 ADMM
iterations of every residence with the state in registers, the voltage rows of a block of 32
iterations are judged by the tree form of R p in one launch (on a second stream for bursts longer
than a block), a failed verdict is rolled back through the four rotating sets of state buffers,
and the max diff of every iteration -- the convergence test -- is folded on the device into the
records the host reads.  The transient's cost, the binding regime (folded chain), binary
residences and the 125 000 x 96 shape of BASELINE config 4 are measured in the same run and
reported as first-class fields beside the headline.

Prints ONE JSON line (rank 0).  `value` = home-QP solves per second, whole job, inputs resident
in HBM, from the MEDIAN of `--bursts` (15) timed regions of exactly `--steps` iterations each (every one
bracketed by barrier + synchronize; all listed in `bursts_ms_per_step`); `value_ev_only` counts the residences
with an EV alone (adoption is stated in config.workload); `value_kkt_presolve`: the same regime with
revs_pdhg_t::polish = 3 (KKT steps from the carried multiplier before PDHG, which then is not entered);
`value_feeder_121144`: the reference's own feeder, 15 iterations, on/off chargers, the oracle timed beside it;
`collective` (N > 1): the block all-reduce's duration beside its block of sweeps.  `roofline`: the sweep kernel against HBM -- algorithmic bytes per launch / average
launch duration from two HIP events around the timed region on its stream / sweep launches in it;
`roofline_valu`: the same kernel against VALU issue (what bounds it since round 3);
`roofline_binding`: the binding regime's launches; `roofline_matvec`: the f64 matrix-core product
R p (feeders that are not radial); `roofline_125k_T96`: config 4's per-GPU shape from the committed PMC passes; `cpu_baseline`: the oracle's vectorised home QP over all
residences on this box's cores (one process per core) plus the operator's voltage product, same
state, same run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Host-side thread pools (OpenBLAS under numpy, OpenMP/MKL under torch) keep spinning for a while
# after every parallel region.  On a box whose CPU share is a cgroup quota (16 CPUs per GPU here,
# 256 visible) two pools of 32-64 spinning threads use the quota up, and the kernel then freezes
# EVERY thread of the process -- the one that feeds the GPU too -- for the rest of the 100 ms
# period: measured as 20-50 ms holes in 5-10 % of the bursts (tools/comm_probe.py).
# So the pools are capped before numpy / torch load (torch.distributed.run sets OMP_NUM_THREADS=1
# itself for N > 1).  Nothing on the hot path runs on these threads.
for _k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "4")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)
F64_MFMA_PEAK_TFLOPS = 78.6    # MI355X datasheet, dense f64 matrix
# VALU issue: shader cycles one SIMD needs per wave64 VALU instruction.  2 = the SIMD-32 rate behind the 157.3 TFLOP/s fp32
# vector peak (MI355X_MICROARCH.md: `v_fma_f32` (wave64) 2 cyc; one wave alone: 4).  Measured on this part by
# tools/probes/valu_rate.hip (s_memtime around every wavefront's loop, first start to last end per SIMD, the wavefronts'
# SIMDs read from HW_ID; raw output: profiles/r05_probe_valu_rate.txt): v_fma_f32 / v_add_f32 with 32 independent chains
# per wavefront sustain 5.0 cycles per instruction with one wavefront per SIMD, 2.5 with two, 2.31-2.48 with four or eight
# (the loop's own branch included) -- NOT 4, which rounds 3-4 priced this kernel against (0.79 then; 0.40 by this);
# v_add_f32_dpp, v_med3_f32, v_pk_fma_f32 and v_fma_f64 run at half that rate (4.4 cycles).
VALU_CYCLES = 2
VALU_CYCLES_SUSTAINED = 2.31                     # best measured: v_fma_f32, 8 wavefronts per SIMD, 16 chains
VALU_PEAK = 1024 * 2.4e9 / VALU_CYCLES           # wave64 VALU instructions per second, 1024 SIMDs at 2.4 GHz
VALU_PEAK_SOURCE = ("2 shader cycles per wave64 VALU instruction per SIMD (SIMD-32; MI355X_MICROARCH.md constants table) x 1024 SIMDs x "
                    "2.4 GHz; profiles/r05_probe_valu_rate.txt (tools/probes/valu_rate.hip) measures 2.31-2.48 cycles sustained "
                    "for v_fma_f32 / v_add_f32 at 4-8 wavefronts per SIMD, 4.4 for DPP adds, v_med3_f32, v_pk_fma_f32, v_fma_f64")
SQ_QUAD = 4     # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md, constants table)
VALU_PEAK = 1024 * 2.4e9 / VALU_CYCLES           # wave64 VALU instructions per second, 1024 SIMDs at 2.4 GHz
VALU_PEAK_SOURCE = ("profiles/r05_probe_valu_rate.txt (tools/probes/valu_rate.hip: cycles per v_fma_f32 / v_add_f32 per SIMD by "
                    "s_memtime at 1 / 2 / 4 / 8 wavefronts per SIMD, 8 and 16 independent chains per wavefront)")


def multi_bytes_per_home(T, pdhg_dual, hist=False, inner=1):
    """Algorithmic HBM bytes of one launch of the multi-iteration sweep per residence
    (revs_agent_step_multi): reads LOAD, P_est, P_sch, G (4 profiles), the
    32-byte record and the node index; writes P_est, P_sch, G (3 profiles), dsq, status and diff
    (one float per inner iteration when the history is kept, else one); the carried PDHG
    multiplier each way.  The call's last launch also writes the prepared estimate (not counted)."""
    b = 4 * 4 * T + 32 + 4 + 3 * 4 * T + 8 + 4 * (inner if hist else 1)
    if pdhg_dual == "full":
        b += 2 * 4 * T
    elif pdhg_dual:
        b += 2 * 4
    return b


def agent_bytes_per_home(T, pdhg_dual, fused=True, recompute=False):
    """Algorithmic HBM bytes of one sweep launch per residence:
    reads LOAD, P_est[k], P_est[k+1], P_sch[k], G[k] (5 profiles) + the 32-byte home record;
    writes P_sch[k+1], G[k+1] (2 profiles) + diff + dsq + status (12 bytes); the PDHG
    multiplier carried across iterations (one float per home each way; one profile each way
    with full_rows); with the next operator home pass folded in (`fused`) the node index
    (4 bytes) and the P_est[k+2] candidate (one profile written); `recompute`: P_est[k+1] is
    recomputed from the state instead of read (one profile less)."""
    b = 5 * 4 * T + 32 + 2 * 4 * T + 12
    if pdhg_dual == "full":
        b += 2 * 4 * T
    elif pdhg_dual:
        b += 2 * 4
    if fused:
        b += 4 + 4 * T
    if recompute:
        b -= 4 * T
    return b


# ---- CPU baseline (oracle; rank 0, N = 1 only) ------------------------------------------
def _cpu_home_chunk(args):
    """Worker (spawned process, numpy only): the oracle's exact relaxed home QP for one block of
    residences, `reps` times; returns the seconds spent solving."""
    cost, load, rec, pe, ps, gm, kappa, reps = args
    sys.path.insert(0, ROOT)
    from oracle import revs_oracle as ro
    oh = ro.homes_from_records(load, rec)
    t0 = time.perf_counter()
    for _ in range(reps):
        ro.home_solve_relaxed(cost, oh, pe, ps, gm, kappa)
    return time.perf_counter() - t0


def cpu_baseline(w, state, budget_s=20.0):
    """The oracle (numpy float64 restatement of lpsolver.py) on the host's cores, on the state
    the GPU run has reached: (i) every residence's QP (`home_solve_relaxed`: the exact optimum
    the PDHG kernel iterates towards), residences split over one process per core; (ii) the
    operator's side of a steady-state iteration -- g0, its node sums and the dense voltage
    product Rn p (numpy BLAS, the parent's capped pool).  value = residences / (t_homes + t_operator)."""
    import multiprocessing as mp
    from oracle import revs_oracle as ro
    pe, ps, gm = (np.asarray(a, np.float64) for a in state)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    procs = max(1, min(cores, 16, w.N // 2000 or 1))
    edges = np.linspace(0, w.N, procs + 1).astype(int)
    # one probe block to size the repetitions (about budget_s of wall time in total)
    t0 = time.perf_counter()
    n_probe = min(w.N, 4000)
    _cpu_home_chunk((w.cost, w.load[:n_probe], w.homes[:n_probe], pe[:n_probe], ps[:n_probe],
                     gm[:n_probe], w.kappa, 1))
    per_home = (time.perf_counter() - t0) / n_probe
    reps = int(np.clip(0.6 * budget_s / max(per_home * w.N / procs, 1e-3), 1, 20))
    jobs = [(w.cost, w.load[a:b], w.homes[a:b], pe[a:b], ps[a:b], gm[a:b], w.kappa, reps)
            for a, b in zip(edges[:-1], edges[1:])]
    ctx = mp.get_context("spawn")                     # children never touch the GPU runtime
    with ctx.Pool(procs) as pool:
        pool.map(_cpu_home_chunk, [(w.cost, w.load[:8], w.homes[:8], pe[:8], ps[:8], gm[:8], w.kappa, 1)] * procs)
        t0 = time.perf_counter()
        pool.map(_cpu_home_chunk, jobs, chunksize=1)
        t_home = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    oreps = 3
    for _ in range(oreps):
        g0 = np.maximum(ro.utility_g0(pe, ps, gm, w.kappa), 0.0)
        p = np.zeros((w.M, w.T))
        np.add.at(p, w.node_of, g0)
        v = w.Rn @ p
        _ = float(v.max())
    t_op = (time.perf_counter() - t0) / oreps
    return {"value": w.N / (t_home + t_op), "unit": "solves/s", "cores": int(procs), "kind": "port",
            "home_qp_solves_per_sec": w.N / t_home, "operator_check_ms": t_op * 1e3,
            "sample": f"one ADMM iteration's work on all {w.N} residences of the same workload at the "
                      f"state the GPU run reached: oracle home QP (exact relaxed optimum, numpy float64) "
                      f"on {procs} processes x {reps} repetitions ({t_home:.2f} s per pass), plus g0, node "
                      f"sums and the dense {w.M}x{w.M}x{w.T} voltage product ({t_op * 1e3:.0f} ms, numpy BLAS)"}


# ---- the reference's own case: lpsolver.solve_ADMM's 15 iterations on the 121144 feeder ----------------
def feeder_121144(torch, with_cpu):
    """The reference's runs on its own feeder (tests/golden/revs_121144.npz: the network, base loads and tariff
    of /root/reference's input files), through the product's call-surface helpers (compute_Rmat, feeder_arrays,
    pack_homes as lpsolver.solve_ADMM uses them): iter_max = 15, on/off chargers (lpsolver.py:92-98, 243) --
    BASELINE config 0's feeder at the stored run's 90 % adoption in community 2 (T = 24) and config 3 (all
    communities, 90 %, T = 96: hourly data held over four 15-minute slots).  Wall time of the 15 iterations on a
    warm engine (second run: code objects loaded), operator work, and -- with_cpu -- the oracle's
    solve_ADMM(util_method="dual") of the same problem timed beside it (CPU baseline of this case)."""
    import networkx as nx
    from revs_admm_amd.engine import AdmmEngine, pack_homes
    from revs_admm_amd.lpsolver import compute_Rmat, feeder_arrays
    z = np.load(os.path.join(ROOT, "tests", "golden", "revs_121144.npz"))
    g = nx.Graph()
    for nid, lab in zip(z["node_id"], z["node_label"]):
        g.add_node(int(nid), label=lab.decode())
    for u, v, r in zip(z["edge_u"], z["edge_v"], z["edge_r"]):
        g.add_edge(int(z["node_id"][u]), int(z["node_id"][v]), r=float(r))
    res = [n for n in g if g.nodes[n]["label"] == "H"]
    nonsub = [n for n in g.nodes if g.nodes[n]["label"] != "S"]
    pos = {n: i for i, n in enumerate(nonsub)}
    R = compute_Rmat(g)
    ri = [pos[n] for n in res]
    R_res = R[np.ix_(ri, ri)]
    feeder = feeder_arrays(g, res)
    row = {int(h): i for i, h in enumerate(z["res_id"])}
    load_res = np.stack([z["LOAD"][row[h]] for h in res])
    out = {}
    for tag, T, ev_ids in (("com2_90pct_T24", 24, z["dis_a90_r4800_ev_homes"]), ("all_communities_90pct_T96", 96, None)):
        rep = T // 24
        if ev_ids is None:
            np.random.seed(1234)                      # revs_fixture.py:174-177
            ev_ids = np.random.choice(z["res_id"], int(0.9 * len(z["res_id"])), replace=False)
        evset = set(int(h) for h in ev_ids)
        ev = np.array([h in evset for h in res])
        load = np.repeat(load_res, rep, axis=1)
        cost = np.repeat(z["tariff_shift6"], rep)
        homes = pack_homes(ev, 4.8, 20.0, 0.2, 11 * rep, 23 * rep)           # (the reference's slot arithmetic unchanged)
        e = AdmmEngine(cost, homes, load, np.arange(len(res)), R_res, kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05,
                       mode="binary", feeder=feeder)
        e.run(15)
        e.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run(15)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row_out = {"ms_15_iterations": dt * 1e3, "home_solves_per_sec": len(res) * 15 / dt, "residences": len(res),
                   "ev_residences": int(ev.sum()), "T": T, "operator_evaluations": int(sum(e.op_iters_hist)),
                   "newton_iterations": [int(h[0]) for h in e.newton_hist],
                   "rows_judged_by": "tree form" if e._tree_eval else "dense f64 product",
                   "fused_launches": bool(e._tree_newton), "tree_recovered_from_matrix": bool(getattr(e, "tree_recovered", False)),
                   "chained_iterations": int(e.chain_hist[0])}
        if T == 96:
            # BASELINE config 3 is named "operator R.p voltage matvec on MFMA": the f64 matrix-core product at THIS shape
            # (M = 1 126 residence rows, T = 96 columns).  The product's own launch, 200 times, by HIP events -- the run
            # above judges its rows by the tree form (rows_judged_by), OperatorOptions(voltage="dense") puts this kernel
            # on the path instead; its MFMA counters: profiles/r05_pmc_mfma.csv (tools/matvec_run.py --config3).
            reps = 200
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e.pnq.uniform_(0.0, 3.0)
            for _ in range(20):
                e._gemm1(e.R64T, e.pnq[2], e.v_sl)
            e0.record()
            for _ in range(reps):
                e._gemm1(e.R64T, e.pnq[2], e.v_sl)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            fl = 2.0 * e.M * e.M * T
            row_out["matvec"] = {"kernel": "gemm_tn_kernel<double> (R p, M x M x T)", "M": int(e.M), "T": T, "bound": "mfma",
                                 "achieved": fl / (ms * 1e-3) / 1e12, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": fl / (ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, "avg_launch_ms": ms,
                                 "matrix_stream_GBs": 8.0 * e.M * e.M / (ms * 1e-3) / 1e9,
                                 "mfma_counters": "profiles/r05_pmc_mfma.csv (rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES ... "
                                                  "of tools/matvec_run.py --config3)"}
        if with_cpu:
            from oracle import revs_oracle as ro
            oh = ro.homes_from_records(load, homes)
            t0 = time.perf_counter()
            ro.solve_ADMM(oh, R_res, np.arange(len(res)), cost, 5.0, 15, 1.03, 0.95, 1.05, mode="binary", util_method="dual")
            row_out["cpu_oracle_ms_15_iterations"] = (time.perf_counter() - t0) * 1e3
        out[tag] = row_out
        del e
        torch.cuda.empty_cache()
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks of this script (this
    process has made no GPU call and makes none), relay rank 0's stdout, exit with the first
    failing rank's code.  Children are separate processes started with Popen -- never exec."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", REVS_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # Watch every child: a rank that dies before the rendezvous would leave the others waiting in
    # init_process_group for the backend's own timeout (10-30 min).  Rank 0's stdout is drained by a
    # thread (its JSON line is long); the first non-zero exit ends the rest.
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc, deadline = 0, time.monotonic() + float(os.environ.get("REVS_BENCH_TIMEOUT", "3000"))
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or all(c is not None for c in codes) or time.monotonic() > deadline:
            rc = bad[0] if bad else (0 if all(c == 0 for c in codes) else 124)
            break
        time.sleep(0.05)
    for p in procs:                       # (exactly the processes started above)
        if p.poll() is None:
            p.terminate()
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10)
    out0 = buf[0] if buf else ""
    for line in (out0 or "").splitlines():       # the JSON line to stdout, library chatter to stderr
        print(line, file=sys.stdout if line.lstrip().startswith("{") else sys.stderr)
    sys.stdout.flush()
    if rc:
        raise SystemExit(rc if rc > 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --homes residences per GPU; strong: --homes in total, split over the GPUs")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks as child processes even for --gpus 1 (rehearsal of the launcher)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only (gloo, no GPU): rank 0 prints the launch topology -- the CPU "
                         "test of the launcher")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks on cuda:0, node sums through the hook communicator over gloo "
                         "(rehearsal of N > 1 on a one-GPU box)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bursts", type=int, default=None,
                    help="the timed region -- exactly --steps steps between two barriers -- is repeated this many "
                         "times back to back; ms_per_step / value are the MEDIAN burst, every burst is listed "
                         "(bursts_ms_per_step).  One burst of 20 steps is a 0.3 ms sample.  Default: 15, but no more "
                         "than fit into ~400 iterations in all (2 at --steps 200): the run reaches the eps-residual "
                         "near iteration 490 and drifts back to the voltage boundary -- another regime, reported as "
                         "value_binding -- a few hundred iterations later")
    ap.add_argument("--adoption", type=float, default=0.5,
                    help="share of residences with an EV (the others' home problem is trivial: p = 0)")
    ap.add_argument("--spinup", type=int, default=30,
                    help="ADMM iterations run (untimed, but reported) before the warm-up so that "
                         "the timed steps start from a mid-run state")
    ap.add_argument("--homes", type=int, default=100_000,
                    help="residences per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--T", type=int, default=24)
    ap.add_argument("--nodes", type=int, default=2048)
    ap.add_argument("--mode", default="pdhg", choices=["pdhg", "relaxed_exact", "binary"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--stress", type=float, default=1.0,
                    help="coordinated-profile voltage / limit of the synthetic feeder")
    ap.add_argument("--voltage", default="auto", choices=["auto", "tree", "dense"],
                    help="steady-state row check: tree form of R p inside the sweep's launch, or "
                         "the dense f64 matrix-core product as its own launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the binding / binary / 125k x 96 regimes (N = 1 only anyway)")
    ap.add_argument("--clock-warm", type=int, default=2000,
                    help="untimed throw-away products enqueued after the warm-up steps so that the "
                         "timed region starts at steady GPU clocks (0: none)")
    ap.add_argument("--eps", type=float, default=1e-4, help="ADMM residual target")
    ap.add_argument("--pdhg-check", type=int, default=None, help="PDHG: convergence test period")
    ap.add_argument("--pdhg-polish", type=int, default=None,
                    help="revs_pdhg_t::polish bits (1: KKT Newton steps behind PDHG, 2: the same steps from the carried "
                         "multiplier before it; default: the library's)")
    ap.add_argument("--lanes", type=int, default=None,
                    help="lanes of a wavefront per residence (revs_pdhg_t::lanes: 16 or 32 at T <= 32; default: the engine's choice)")
    ap.add_argument("--op-kadd", type=int, default=None,
                    help="operator: violated rows admitted to a slot's model per Newton iteration")
    ap.add_argument("--stream-block", type=int, default=None,
                    help="sharded steady state: iterations per all-reduce / verdict launch (default 32)")
    ap.add_argument("--stream-inner", type=int, default=None,
                    help="steady state: ADMM iterations per sweep launch (default: OperatorOptions.stream_inner)")
    ap.add_argument("--no-stream-overlap", action="store_true",
                    help="sharded steady state: collective and verdicts on the compute stream")
    ap.add_argument("--no-converge", action="store_true",
                    help="skip the untimed run to the eps-residual (profiling runs)")
    args = ap.parse_args()

    if args.bursts is None:
        args.bursts = int(np.clip(400 // max(args.steps, 1), 1, 15))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        return spawn_ranks(args.gpus)        # before anything touches the GPU

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}")
    if args.dry_run:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank), float(local), 1.0], dtype=torch.float64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "sum_ranks": t[0].item(),
                              "sum_local_ranks": t[1].item(), "ranks_seen": t[2].item(),
                              "scaling": args.scaling,
                              "launcher": "self" if os.environ.get("REVS_BENCH_CHILD") else "external"}),
                  flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    group = None
    if world > 1 or os.environ.get("REVS_FORCE_GROUP") or os.environ.get("REVS_BENCH_CHILD"):
        # (a 1-rank group rehearses the N > 1 code on one GPU)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1")
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))
        group = dist.group.WORLD

    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def build(homes, T, mode, stress, voltage, lanes_for=None, polish=None):
        n_total = homes * world if args.scaling == "weak" else homes
        w = make_workload(n_total, T, n_nodes=args.nodes, seed=args.seed, adoption=args.adoption,
                          binary_feasible=(mode == "binary"), stress=stress)
        lo, hi = w.shard(rank, world)
        counts = np.bincount(w.node_of, minlength=w.M)
        opts = OperatorOptions(voltage=voltage)
        if args.op_kadd is not None:
            opts.newton_kadd = args.op_kadd
        if args.stream_block is not None:
            opts.stream_block = args.stream_block
        if args.stream_inner is not None:
            opts.stream_inner = args.stream_inner
        opts.stream_overlap = not args.no_stream_overlap
        # few residences per GPU (BASELINE config 2 over eight GPUs: 12 500 each): 16 lanes x 2 slots per residence
        # (revs_pdhg_t::lanes; measured at 12 500 / 25 000 / 50 000 x 24: 0.0044 / 0.0052 / 0.0069 ms per iteration
        # against 0.0049 / 0.0040 / 0.0054 with the default 8 x 3 -- the per-iteration chain of a lone wavefront is
        # its cross-lane reductions, not its per-slot work, so the wide shapes pay only below ~16 000 residences)
        lanes = args.lanes if args.lanes is not None else (16 if (T <= 32 and (lanes_for or hi - lo) <= 16000) else 0)
        build.lanes = lanes
        eng = AdmmEngine(w.cost, w.homes[lo:hi], w.load[lo:hi], w.node_of[lo:hi], w.Rn, kappa=w.kappa,
                         vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode=mode, device=dev, group=group,
                         node_counts=counts, op=opts, feeder=w.feeder,
                         pdhg=({k: v for k, v in (("check", args.pdhg_check), ("polish", args.pdhg_polish if polish is None else polish), ("lanes", lanes or None)) if v is not None} or None))
        return w, eng, (lo, hi)

    def timed_steps(eng, k):
        """K steps, nothing else on the stream; wall clock between two barriers (MAX over ranks)
        and two HIP events on the kernels' stream around the same K steps -- recorded by the
        library itself right before the first and right after the last launch of the streaming
        loop (revs_plan_stream_timing), by torch around run_steps on the other paths."""
        import ctypes as C
        native = eng._plan is not None and eng._tree is not None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if native:
            eng.lib.revs_plan_stream_timing(eng._plan, 1)
        barrier()
        t0 = time.perf_counter()
        if not native:
            e0.record()
        eng.run_steps(k)
        if not native:
            e1.record()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.share_gpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        ms = C.c_double()
        if native and eng.lib.revs_plan_stream_elapsed_ms(eng._plan, C.addressof(ms)) == 0:
            dt_ev = ms.value * 1e-3
            timed_steps.launches = int(eng.lib.revs_plan_stream_launches(eng._plan))
        else:                       # (no streaming burst in the region: nothing to price)
            dt_ev = e0.elapsed_time(e1) * 1e-3 if not native else float("nan")
        cm, bm, ci = C.c_double(), C.c_double(), C.c_int32()
        if native and world > 1 and eng.lib.revs_plan_collective_ms(eng._plan, C.addressof(cm), C.addressof(bm),
                                                                    C.addressof(ci)) == 0:
            timed_steps.collective = (cm.value, bm.value, ci.value)
        if native:
            eng.lib.revs_plan_stream_timing(eng._plan, 0)
        return dt, dt_ev

    def clock_warm(eng):
        for _ in range(args.clock_warm):
            eng._gemm1(eng.R64T, eng.pnq[2], eng.v_sl)

    # The first launch of every kernel loads its code object (4 - 5 ms in all, once per process): a small throw-away
    # engine of the same shape walks through the transient and into the steady state first, so that the spin-up below
    # prices the iterations, not the loader (tools/transient_times.py: 8.5 ms for the first engine's 30 iterations, 3.5 ms
    # for the second one's in the same process).
    if world == 1 and not os.environ.get("REVS_BENCH_COLD"):
        args_homes = args.homes
        wq, eq, _ = build(min(args.homes, 4096), args.T, args.mode, args.stress, args.voltage, lanes_for=args.homes)
        for _ in range(14):
            eq.step(write_sc=False)
        eq.run_steps(40)
        torch.cuda.synchronize()
        del wq, eq
        torch.cuda.empty_cache()
        assert args.homes == args_homes
    w, eng, (lo, hi) = build(args.homes, args.T, args.mode, args.stress, args.voltage)
    n_total, n_local = w.N, hi - lo
    barrier()       # ranks leave the set-up together (a step waits for every rank's all-reduce)

    # Spin-up: the transient -- every charger jumps to the cheapest slots, voltage rows bind and
    # residences are clamped; run outside the timed region, reported under breakdown.transient.
    spin_ms = []
    for _ in range(args.spinup):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.step(write_sc=False)
        torch.cuda.synchronize()
        spin_ms.append((time.perf_counter() - t1) * 1e3)
    spin_inner = list(eng.op_iters_hist)
    spin_paths = "".join(p[0] for p in eng.op_path_hist)
    eng.run_steps(args.warmup)
    # The spin-up synchronises after every iteration and W steps are a fraction of a millisecond:
    # keep the GPU busy for ~25 ms with untimed throw-away work (voltage products on scratch
    # operands, no ADMM state touched) so that the timed region starts at steady clocks.
    clock_warm(eng)
    spec0 = list(eng.spec_hist)
    inner0 = len(eng.op_iters_hist)
    # ---- the timed region: exactly K steps between two barriers, `--bursts` times; the median burst counts ----
    burst_rows = []
    for _ in range(max(1, args.bursts)):
        k0 = eng.spec_hist[0]
        timed_steps.launches = 0
        b_dt, b_ev = timed_steps(eng, args.steps)
        burst_rows.append((b_dt, b_ev, getattr(timed_steps, "launches", 0) or args.steps, eng.spec_hist[0] - k0))
    order = sorted(range(len(burst_rows)), key=lambda i: burst_rows[i][0])
    dt, dt_ev, n_launch, _ = burst_rows[order[len(order) // 2]]       # (residence-sweep launches in that burst)
    bursts_ms = [r[0] / args.steps * 1e3 for r in burst_rows]
    kept = eng.spec_hist[0] - spec0[0]
    streamed = bool(eng._tree is not None and all(r[3] == args.steps for r in burst_rows))
    inner = eng.op_iters_hist[inner0:]
    rp, rd, dmax, conv = eng.residuals(args.eps)
    st = eng.status.cpu().numpy()
    pdhg_it = float((st >> 8).sum()) / max(1, int((w.homes[lo:hi]["ev"] != 0).sum())) if args.mode == "pdhg" else None

    # the same regime with the rows judged by the dense f64 product on the matrix cores (its own
    # launch before every sweep, verdict read by the host): what the tree form replaces.  (A short stretch: the
    # run reaches the eps-residual near iteration 490 and the voltage boundary a few hundred iterations later.)
    def dense_comparison():
        if not (eng._tree is not None and eng._plan is not None and world == 1):
            return None
        import ctypes as C
        k = min(args.steps, 50)
        tree, eng._tree = eng._tree, None                    # (the engine and its native plan both drop the tree)
        eng.lib.revs_plan_set_tree(eng._plan, None)
        eng.run_steps(args.warmup)
        d_dt, _ = timed_steps(eng, k)
        eng._tree = tree
        eng.lib.revs_plan_set_tree(eng._plan, C.byref(tree))
        return {"ms_per_step": d_dt / k * 1e3, "value": n_total * k / d_dt, "steps": k}

    dense_first = eng.iteration + args.warmup + min(args.steps, 50) < 440       # (else: behind the convergence count)
    dense = dense_comparison() if dense_first else None

    # the f64 matrix-core product of the operator's Newton path: R p (M x M x T)
    reps = 200
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng._gemm1(eng.R64T, eng.pnq[2], eng.v_sl)
    e1.record()
    torch.cuda.synchronize()
    gemm_ms = e0.elapsed_time(e1) / reps
    gemm_flops = 2.0 * eng.M * eng.M * args.T

    # how many ADMM iterations until max_h diff stays at the eps-residual (continues the run)
    # -- from the max diff every streamed iteration leaves in its record (folded on the device by the
    # sweeps and the verdict launches: no extra launch, no read-back of diff); iterations that ran
    # outside the streaming loop are judged by revs_residual_finalize as before
    iters_to_eps = None
    if args.mode != "binary" and not args.no_converge:
        # AdmmEngine.run's rule: max diff <= eps for 8 iterations in a row (an iteration outside the
        # streaming loop -- rows binding -- has no record and breaks the stretch)
        seen, good = 0, 0
        while eng.iteration < 1000 and iters_to_eps is None:
            while seen < eng.iteration and iters_to_eps is None:
                seen += 1
                good = good + 1 if eng.max_diff.get(seen, np.inf) <= args.eps else 0
                if good >= 8:
                    iters_to_eps = seen - 7
            if iters_to_eps is None:
                eng.run_steps(64)
        rp, rd, dmax, conv = eng.residuals(args.eps)
    if not dense_first:
        dense = dense_comparison()
    state = eng.get_state() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    # ---- the other regimes, first-class (one GPU) ----
    extras = {}
    if world == 1 and not args.no_extras:
        def regime(homes, T, mode, stress, spin, steps, nblocks=3, polish=None):
            w2, e2, _ = build(homes, T, mode, stress, args.voltage, polish=polish)
            t1 = time.perf_counter()
            e2.run_steps(spin)
            torch.cuda.synchronize()
            spin_s = time.perf_counter() - t1
            clock_warm(e2)
            s0, c0 = list(e2.spec_hist), list(e2.chain_hist)
            # three blocks of `steps`; the median block is the regime's figure (these regimes have
            # the host in every iteration: a host thread frozen by the box's CPU quota cost a block
            # ~10 ms once in a while before the thread pools were capped, see the top of this
            # file -- all three blocks are still listed)
            blocks = [timed_steps(e2, steps)[0] / steps * 1e3 for _ in range(nblocks)]
            d2 = float(np.median(blocks)) * 1e-3 * steps
            out = {"value": homes * steps / d2, "unit": "solves/s", "ms_per_step": d2 / steps * 1e3,
                   "blocks_ms_per_step": blocks,
                   "homes": homes, "T": T, "home_solver": mode, "stress": stress,
                   "first_iterations": spin, "first_iterations_ms": spin_s * 1e3,
                   "steady_state_steps_kept": e2.spec_hist[0] - s0[0],
                   "chained_newton_steps": e2.chain_hist[0] - c0[0],
                   "operator_evaluations_per_step": float(np.mean(e2.op_iters_hist[-3 * steps:]))}
            if mode == "pdhg":
                st2 = e2.status.cpu().numpy() >> 8
                out["pdhg_passes_mean_over_ev_residences"] = float(st2.sum()) / max(1, int((w2.homes["ev"] != 0).sum()))
            del e2, w2
            torch.cuda.empty_cache()
            return out
        # rows that keep binding: one Newton iteration per ADMM iteration from warm multipliers
        extras["value_binding"] = regime(args.homes, args.T, "pdhg", 1.3, 60, 100)
        # the reference's own home model: binary chargers (MIQP solved exactly by ranking)
        extras["value_binary"] = regime(args.homes, args.T, "binary", args.stress, 40, 100)
        # BASELINE config 4's per-GPU shape
        if (args.homes, args.T) == (100_000, 24):
            extras["value_125k_T96"] = regime(125_000, 96, "pdhg", args.stress, 40, 100)
            # ... and BASELINE config 4 at its stated size, all of it on this one GPU (~6.5 GB of state)
            extras["value_1M_T96"] = regime(1_000_000, 96, "pdhg", args.stress, 40, 100)
        # the headline regime with revs_pdhg_t::polish = 3: the KKT steps run from the carried multiplier BEFORE
        # PDHG too, and in this regime they settle every residence -- PDHG is the fallback that is not entered
        # (pdhg_passes_mean 0).  Reported beside the headline, which keeps PDHG in every solve (polish = 1).
        if args.mode == "pdhg" and args.pdhg_polish is None:
            extras["value_kkt_presolve"] = regime(args.homes, args.T, "pdhg", args.stress, args.spinup + args.warmup,
                                                  args.steps, nblocks=len(burst_rows), polish=3)
        # the reference's own feeder and iteration count
        extras["value_feeder_121144"] = feeder_121144(torch, not args.no_cpu_baseline)
        extras["roofline_matvec_config3"] = extras["value_feeder_121144"]["all_communities_90pct_T96"].pop("matvec", None)

    # Time to the eps-residual, measured: ONE AdmmEngine.run(iter_max = 1000, eps) from the zero state of
    # lpsolver.py:244-246 on a fresh engine -- transient, streaming steady state, the stopping test (max_h diff <= eps
    # for 8 iterations in a row, on the records the launches fold on the device) and the last iteration that writes the
    # schedules -- wall clock between two synchronises, median of 5 fresh engines; and the reference's own stopping
    # rule (iter_max = 15, on/off chargers: lpsolver.py:243, 254) the same way.  history=False: the per-residence diff of
    # every iteration (lpsolver.solve_ADMM's return value: 200 MB at 500 iterations) is not carried to the host; the
    # run WITH it is timed beside.
    tte = None
    if world == 1 and not args.no_extras and not args.no_converge:
        def fresh_runs(mode, iter_max, eps, reps, history=False, polish=None):
            ms, its, conv = [], [], []
            for _ in range(reps):
                w3, e3, _ = build(args.homes, args.T, mode, args.stress, args.voltage, polish=polish)
                clock_warm(e3)               # (building the workload left the GPU idle for ~0.5 s: the run is timed at steady clocks, like the steps)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                r = e3.run(iter_max, eps=eps, history=history)
                torch.cuda.synchronize()
                ms.append((time.perf_counter() - t1) * 1e3)
                its.append(int(r if not history else len(r)))
                conv.append(e3.converged_at)
                del e3, w3, r
                torch.cuda.empty_cache()
            i = int(np.argsort(ms)[len(ms) // 2])
            return {"ms": ms[i], "iterations": its[i], "converged_at": conv[i], "all_ms": ms, "home_solver": mode}
        tte = {"pdhg_eps": fresh_runs("pdhg", 1000, args.eps, 5) if args.mode == "pdhg" else None,
               "pdhg_eps_with_diff_history": fresh_runs("pdhg", 1000, args.eps, 3, history=True) if args.mode == "pdhg" else None,
               # (revs_pdhg_t::polish = 3, as value_kkt_presolve: the KKT steps settle a residence before PDHG is entered)
               "pdhg_eps_kkt_presolve": (fresh_runs("pdhg", 1000, args.eps, 3, polish=3)
                                         if args.mode == "pdhg" and args.pdhg_polish is None else None),
               "reference_rule_binary_15_iterations": fresh_runs("binary", 15, None, 5),
               "reference_rule_binary_15_iterations_with_diff_history": fresh_runs("binary", 15, None, 3, history=True)}

    # The same first iterations as ONE call (what a run does: AdmmEngine.run / run_steps -- the iterations behind the
    # transient stream, 32 to a launch, instead of being issued and waited for one by one as the spin-up above does to
    # time each): the engine is set back to iteration 0 and walks them again.
    one_call_ms = None
    if world == 1 and args.spinup > 0:
        eng.reset()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.run_steps(args.spinup)
        torch.cuda.synchronize()
        one_call_ms = (time.perf_counter() - t1) * 1e3

    if rank == 0:
        warm = (None if eng.pdhg_dual is None else
                ("full" if eng.pdhg_dual.dim() == 2 else "scalar"))
        rec = bool(getattr(eng, "recompute_pe_new", False))
        inner = int(getattr(eng, "_inner", 1)) if eng._block else 1
        if eng._block:
            bph = multi_bytes_per_home(args.T, warm, hist=False, inner=inner)
        else:
            bph = agent_bytes_per_home(args.T, warm, fused=True, recompute=rec)
        bytes_per_launch = bph * n_local
        launch_ms = dt_ev / n_launch * 1e3 if streamed else None
        ach = bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms else None
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "agent_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if (tj.get("homes") == n_local and tj.get("T") == args.T
                        and tj.get("mode") == args.mode
                        and tj.get("algorithmic_bytes_per_launch") == bytes_per_launch):
                    traffic, traffic_src = tj["hbm_bytes_per_launch"], tj.get("source")
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
            "bursts": len(burst_rows),
            "bursts_ms_per_step": bursts_ms,
            "ms_per_step_rule": "median of `bursts` timed regions of exactly `steps` steps each (barrier + synchronize "
                                "on both sides of every one), run back to back; value = residences / that",
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic {n_local} homes/GPU x T={args.T} ("
                            + (f"weak scaling: {args.homes} per GPU" if args.scaling == "weak" else
                               f"strong scaling: {n_total} in total over {world} GPU(s), BASELINE config 2's shape")
                            + f") box+SOC home QP, EV adoption {args.adoption:g} (the other residences' "
                            "problem is p = 0: value_ev_only counts the EV residences alone), "
                            f"{args.nodes}-node radial feeder, one ADMM iteration per step (voltage "
                            "rows of the operator's estimate + all home QPs + dual update + "
                            f"residual terms), timed from a state {args.spinup + args.warmup} ADMM "
                            "iterations into the run; the transient before it is in "
                            "breakdown.transient, the binding / binary / 125k x 96 regimes in "
                            "value_binding / value_binary / value_125k_T96",
                "homes_per_gpu": n_local, "homes_total": n_total, "T": args.T,
                "nodes": args.nodes, "home_solver": args.mode, "kappa": w.kappa,
                "lanes_per_residence": int(eng.pdhg.lanes) or "default (8 x 3 slots at T = 24)",
                "adoption": args.adoption, "ev_residences_total": int((w.homes["ev"] != 0).sum()),
                "operator_dtype": "f64", "parallelism": f"homes sharded x{world}, nodes replicated, "
                                                        "node sums all-reduced"
                + (f", {eng._block} iterations per collective"
                   + (" on a second stream" if eng.op.stream_overlap else "") if eng._block else
                   ", one collective per iteration"),
                "voltage_rows": ("tree form of R p inside the sweep's launch" if streamed else
                                 "dense f64 product R p on the matrix cores"),
                "launches_per_step": (n_launch / args.steps) if streamed else 2,
                "iterations_per_sweep_launch": inner if streamed else 1,
                "iterations_per_timed_launch": (args.steps / n_launch) if streamed else 1,
                "recompute_pe_new": rec,
                "collective": (None if world == 1 and group is None else
                               ("hook communicator over gloo, all ranks on cuda:0 (rehearsal)" if args.share_gpu
                                else "library-owned RCCL communicator" if eng._comm else "torch.distributed")),
                "launcher": ("bench.py spawned the ranks itself" if os.environ.get("REVS_BENCH_CHILD")
                             else "external (torch.distributed.run)" if world > 1 else "single process"),
                "clock_warmup_products": args.clock_warm,
            },
            # N > 1: the all-reduce of a block's node sums (events around it on the stream it is issued on) beside that
            # block's sweep launches: the collective costs the step nothing while it is the shorter of the two
            "collective": (None if not getattr(timed_steps, "collective", None) else
                           {"collective_ms_per_block": timed_steps.collective[0],
                            "block_sweeps_ms": timed_steps.collective[1],
                            "iterations_in_that_block": timed_steps.collective[2],
                            "bytes": int(timed_steps.collective[2]) * (eng.M * args.T + 64 * world) * 8,
                            "bytes_per_iteration": (eng.M * args.T + 64 * world) * 8,
                            # (a T-double screen of the ranks' row maxima in front of it was evaluated on 8 logical shards
                            # of this workload and dropped: profiles/r05_screen_bound.txt)
                            "screened": False,
                            "hidden": bool(eng.op.stream_overlap and timed_steps.collective[0] < timed_steps.collective[1])}),
            "roofline": {
                "kernel": ("agent_step_kernel<MULTI> (%d ADMM iterations of every residence per launch: home QP "
                           "sweeps + dual updates + next operator home passes, state in registers; voltage "
                           "rows judged by blocks of %d iterations in their own launch)" % (inner, eng._block)
                           if streamed and eng._block else
                           "agent_step_kernel (home QP sweep + dual update + next operator home pass"
                           + (" + voltage rows of T slots in its first T workgroups)" if streamed else ")")),
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS if ach else None,
                # PMC counters are collected in separate rocprofv3 passes of this command and
                # committed under profiles/; NOT measured inside this run
                "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_home": bph, "bytes_per_launch": bytes_per_launch,
                # (the timed launches carry steps / n_launch iterations each: 20 in the driver's run)
                "bytes_per_home_per_iteration": bph / max(args.steps / n_launch, 1.0),
                "sweep_launches_timed": n_launch,
                # two HIP events around the timed region on the launches' stream / number of sweep
                # launches in it: the launch duration INCLUDING the inter-kernel boundaries and the
                # block verdict launches that share the stream (rocprofv3's kernel time is shorter)
                "avg_launch_ms": launch_ms,
                "frac_of_achievable_copy_rate": (ach / 6290.0) if ach else None,   # 6.29 TB/s: float4 copy
                "pdhg_iters_mean": pdhg_it,
            },
            "roofline_matvec": {
                "kernel": "gemm_tn_kernel<double> (voltage rows R.p of the operator's Newton path, M x M x T)",
                "bound": "mfma", "achieved": gemm_flops / (gemm_ms * 1e-3) / 1e12,
                "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": gemm_flops / (gemm_ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                "avg_launch_ms": gemm_ms,
                # 6 flop per byte of R at T = 24: the product streams the matrix
                "matrix_stream_GBs": 8.0 * eng.M * eng.M / (gemm_ms * 1e-3) / 1e9,
                "matrix_stream_frac_of_hbm_peak": 8.0 * eng.M * eng.M / (gemm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "on_path": "since round 3 only for feeders given as a matrix, and in the Newton evaluations of "
                           "feeders beyond the tree form's 2048 nodes: the steady state, the Newton evaluations and "
                           "the binding chain judge their rows by the tree form of R p",
                "mfma_counters": "profiles/r05_pmc_mfma.csv (rocprofv3 --pmc of tools/matvec_run.py: 131 072 v_mfma_f64_16x16x4 per launch, "
                                 "SQ_VALU_MFMA_BUSY_CYCLES 8 388 608 = 8 192 per SIMD of the launch's ~32 000 cycles: 25 %; x 24/32 useful tile columns)",
            },
            # the same timed region counted on the residences that have a QP to solve
            "value_ev_only": float((w.homes["ev"] != 0).sum()) * args.steps / dt,
            "value_dense_product_path": dense,
            "breakdown": {
                "operator_inner_iters_mean": float(np.mean(inner)) if inner else 0.0,
                "operator_path": eng.op_path_hist[-1],
                "steady_state_steps_kept_discarded": [kept, eng.spec_hist[1] - spec0[1]],
                "operator_voltage_rows": int(eng.M * args.T),
                "admm_residual_primal": rp, "admm_residual_dual": rd, "admm_max_diff": dmax,
                "admm_iters_to_eps": iters_to_eps, "eps": args.eps,
                "admm_iters_to_eps_rule": "first of 8 consecutive iterations with max_h diff[h] <= eps, from the "
                                          "records the streaming launches fold on the device",
                # whole-run view: transient + steady iterations of a 400-iteration run
                "amortized_400_iterations_solves_per_sec":
                    (n_total * 400 / ((np.sum(spin_ms) + (400 - args.spinup) * dt / args.steps * 1e3)
                                      * 1e-3)) if spin_ms and args.spinup <= 400 else None,
                "transient": {
                    "spinup_iterations": args.spinup,
                    "ms_per_step_mean": float(np.mean(spin_ms)) if spin_ms else None,
                    "ms_per_step_max": float(np.max(spin_ms)) if spin_ms else None,
                    "ms_total": float(np.sum(spin_ms)) if spin_ms else None,
                    "timed": "every iteration issued and waited for on its own (step + synchronize)",
                    "one_call_ms_total": one_call_ms,
                    "one_call_ms_per_step_mean": (one_call_ms / args.spinup) if one_call_ms is not None else None,
                    "one_call": "the same iterations from iteration 0 again as ONE run_steps call (the iterations behind "
                                "the transient stream)",
                    "operator_inner_iters": spin_inner,
                    "operator_paths": spin_paths,
                },
            },
        }
        # VALU issue: the multi-iteration sweep is bound by instruction issue, not by HBM (its bytes per
        # ADMM iteration are 1/8 of the one-iteration sweep's).  Instructions per launch from the committed
        # SQ counter pass (profiles/agent_traffic.json, same workload), duration live (avg_launch_ms).
        # Peak: VALU_PEAK above (one wave64 VALU instruction per 2 cycles per SIMD; profiles/r05_probe_valu_rate.txt).
        valu = None
        try:
            tj = json.load(open(tpath))
            c = tj.get("sq_counters_per_launch", {})
            if (tj.get("homes") == n_local and tj.get("T") == args.T and tj.get("mode") == args.mode
                    and tj.get("iterations_per_launch") == inner and c.get("SQ_INSTS_VALU") and launch_ms):
                peak = VALU_PEAK
                # (the counters are per launch of `inner` iterations; the timed launches carry
                # steps / n_launch of them each -- 20 in the driver's run)
                it_launch = args.steps / n_launch
                ach_v = c["SQ_INSTS_VALU"] * (it_launch / inner) / (launch_ms * 1e-3)
                valu = {"kernel": "agent_step_kernel<MULTI>", "bound": "valu-issue", "achieved": ach_v / 1e9,
                        "peak": peak / 1e9, "peak_source": VALU_PEAK_SOURCE,
                        "unit": "G wave64 VALU instructions/s", "frac": ach_v / peak,
                        "frac_of_measured_sustained_rate": ach_v / (1024 * 2.4e9 / VALU_CYCLES_SUSTAINED),
                        "valu_instructions_per_launch": c["SQ_INSTS_VALU"] * (it_launch / inner),
                        "iterations_per_timed_launch": it_launch,
                        "valu_instructions_per_wavefront_and_iteration": c["SQ_INSTS_VALU"] / max(c.get("SQ_WAVES", 1), 1) / inner,
                        "salu_instructions_per_launch": (c["SQ_INSTS_SALU"] * (it_launch / inner) if c.get("SQ_INSTS_SALU") else None),
                        "valu_busy_share_of_launch": (c["SQ_ACTIVE_INST_VALU"] * (it_launch / inner) * SQ_QUAD / 1024 / (launch_ms * 1e-3 * 2.4e9)
                                                      if c.get("SQ_ACTIVE_INST_VALU") else None),
                        "source": tj.get("source")}
        except Exception:
            valu = None
        out["roofline_valu"] = valu
        # the binding regime's launches (rocprofv3 kernel stats committed under profiles/, same command as
        # tools/regime_run.py): the folded chain's sweep against HBM, and the operator launch, which is a
        # latency chain of 2 T workgroups, not a bandwidth or arithmetic kernel
        rb = None
        try:
            import csv
            import glob
            bfile = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_binding_kernel_stats.csv")))[-1]
            rows = list(csv.DictReader(open(bfile)))
            sw = next(r for r in rows if "agent_step_kernel" in r["Name"] and r["Name"].rstrip().endswith("true>(revs::AgentArgs)")
                      and ", false, true>" in r["Name"])
            kv = next(r for r in rows if "op_chain_kv_kernel" in r["Name"])
            if (args.homes, args.T) == (100_000, 24) and "value_binding" in extras:
                b7 = (4 * 4 * args.T + 32 + 4 + 3 * 4 * args.T + 12 + 8) * 100_000
                t_sw, t_kv = float(sw["AverageNs"]) * 1e-9, float(kv["AverageNs"]) * 1e-9
                rb = {"kernel": "agent_step_kernel<CHAIN> (residence sweep with the operator's answer for the trial "
                                "formed inside and both evaluations' node sums folded in)",
                      "bound": "hbm", "achieved": b7 / t_sw / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": b7 / t_sw / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": t_sw * 1e3, "bytes_per_launch": b7,
                      "operator_launch": {"kernel": "op_chain_kv_kernel (2 T workgroups: trial verdict + next "
                                                    "iteration's rows / selection / model / step / shifts)",
                                          "avg_launch_ms": t_kv * 1e3, "bound": "latency (one workgroup per slot)"},
                      "ms_per_step_live": extras["value_binding"]["ms_per_step"],
                      "source": f"profiles/{os.path.basename(bfile)} (rocprofv3 --kernel-trace --stats of "
                                "tools/regime_run.py --regime binding); ms_per_step_live from this run"}
        except Exception:
            rb = None
        out["roofline_binding"] = rb
        # BASELINE config 4: what bounds the T = 96 sweep at its per-GPU shape (125 000 x 96) and at the config's whole size
        # on ONE GPU (1 000 000 x 96: the residences' state no longer fits the Infinity Cache) -- counters from the
        # committed --pmc passes of tools/regime_run.py, duration live from this run's value_125k_T96 / value_1M_T96
        def t96_roofline(tfile, key, homes):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", tfile)))
                c = tj["sq_counters_per_launch"]
                if not (key in extras and tj["homes"] == homes and tj["T"] == 96):
                    return None
                it, us = tj["iterations_per_launch"], tj["avg_launch_us_profiled"]
                peak = VALU_PEAK
                return {"kernel": tj["kernel"] + f" ({it} ADMM iterations of {homes} residences x 96 slots per launch)",
                        "bound": "valu-issue", "achieved": c["SQ_INSTS_VALU"] / (us * 1e-6) / 1e9, "peak": peak / 1e9,
                        "peak_source": VALU_PEAK_SOURCE,
                        "unit": "G wave64 VALU instructions/s", "frac": c["SQ_INSTS_VALU"] / (us * 1e-6) / peak,
                        "valu_busy_share_of_launch": c["SQ_ACTIVE_INST_VALU"] * SQ_QUAD / 1024 / (us * 1e-6 * 2.4e9),
                        "valu_instructions_per_wavefront_and_iteration": c["SQ_INSTS_VALU"] / c["SQ_WAVES"] / it,
                        "salu_instructions_per_wavefront_and_iteration": c["SQ_INSTS_SALU"] / c["SQ_WAVES"] / it,
                        "hbm": {"bound": "hbm", "achieved": tj["algorithmic_bytes_per_launch"] / (us * 1e-6) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": tj["algorithmic_bytes_per_launch"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "traffic": tj["hbm_bytes_per_launch"], "bytes_per_launch": tj["algorithmic_bytes_per_launch"]},
                        "avg_launch_ms_profiled": us * 1e-3, "ms_per_step_live": extras[key]["ms_per_step"],
                        "source": tj["source"]}
            except Exception:
                return None
        out["roofline_125k_T96"] = t96_roofline("t96_traffic.json", "value_125k_T96", 125_000)
        out["roofline_1M_T96"] = t96_roofline("t96_1m_traffic.json", "value_1M_T96", 1_000_000)
        out.update(extras)
        out["cpu_baseline"] = cpu_baseline(w, state) if state is not None else None
        if tte is not None:
            cpu_it_s = (n_total / out["cpu_baseline"]["value"]) if out.get("cpu_baseline") else None
            a, b = tte["pdhg_eps"], tte["reference_rule_binary_15_iterations"]
            out["time_to_eps"] = {
                "time_to_eps_ms": a["ms"] if a else None, "iterations": a["iterations"] if a else None,
                "converged_at": a["converged_at"] if a else None, "eps": args.eps, "runs_ms": a["all_ms"] if a else None,
                "rule": "ONE AdmmEngine.run(iter_max=1000, eps, history=False) from the zero state on a fresh engine (GPU clocks "
                        "warmed by throw-away products first, as before the timed steps), wall clock between two synchronises, "
                        "median of 5 engines; stops after max_h diff <= eps has held for 8 iterations "
                        "(+ at most the rest of a 64-iteration burst) and one more iteration that writes the schedules",
                "with_per_residence_diff_history_ms": (tte["pdhg_eps_with_diff_history"] or {}).get("ms"),
                "kkt_presolve": ({"ms": tte["pdhg_eps_kkt_presolve"]["ms"], "iterations": tte["pdhg_eps_kkt_presolve"]["iterations"],
                                  "runs_ms": tte["pdhg_eps_kkt_presolve"]["all_ms"],
                                  "note": "the same run with revs_pdhg_t::polish = 3 (value_kkt_presolve's option; not the default)"}
                                 if tte.get("pdhg_eps_kkt_presolve") else None),
                "cpu_time_to_eps_s_projected": (a["iterations"] * cpu_it_s) if (a and cpu_it_s) else None,
                "cpu_projection": "iterations x the cpu_baseline's time for one iteration's work (steady-state work; the "
                                  "transient's operator QPs would come on top)",
                "reference_rule": {"ms": b["ms"], "iterations": b["iterations"], "home_solver": "binary", "runs_ms": b["all_ms"],
                                   "with_per_residence_diff_history_ms":
                                       tte["reference_rule_binary_15_iterations_with_diff_history"]["ms"],
                                   "rule": "the reference's own stopping rule: iter_max = 15 iterations, on/off chargers "
                                           "(lpsolver.py:243, 254), AdmmEngine.run(15) from the zero state",
                                   "cpu_s_projected": (15 * cpu_it_s) if cpu_it_s else None},
            }
        print(json.dumps(out), flush=True)
    if group is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
