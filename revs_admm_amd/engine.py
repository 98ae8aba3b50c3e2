"""Array-level ADMM engine: owns the device buffers and drives librevs_admm.so.

One `AdmmEngine` holds one rank's shard of residences (all of them on a single
GPU) plus a replica of the node-space operator state.  Mirrors the data flow of
`lpsolver.solve_ADMM` (reference lpsolver.py:242-290); the dict-level call
surface lives in `revs_admm_amd.lpsolver`.

PyTorch is used for device memory, streams and (multi-GPU) the RCCL all-reduce of
the node aggregate -- every number is computed by the HIP kernels behind the C
ABI.  There is no CPU path: constructing an engine without a GPU or without the
built library raises.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import HOME_DTYPE, MODES, PDHG, check, ptr

SOC_TARGET, SOC_MAX, _SOC_TOL = 0.9, 1.0, 1e-9


def pack_homes(ev, rating, capacity, initial, start, end) -> np.ndarray:
    """Per-residence records (revs_home_t).  nmin/nmax -- the slot counts the SOC
    rows of lpsolver.py:101-109 allow -- are computed here in double so that the
    float kernels never decide a borderline case."""
    ev = np.asarray(ev, bool)
    n = len(ev)
    rating = np.broadcast_to(np.asarray(rating, float), (n,))
    capacity = np.broadcast_to(np.asarray(capacity, float), (n,))
    initial = np.broadcast_to(np.asarray(initial, float), (n,))
    rec = np.zeros(n, HOME_DTYPE)
    with np.errstate(divide="ignore", invalid="ignore"):
        per = np.where(ev, rating / np.where(capacity != 0, capacity, 1.0), 1.0)
        per = np.where(per > 0, per, 1.0)
    nmin = np.ceil((np.maximum(SOC_TARGET, initial) - initial) / per - _SOC_TOL)
    nmax = np.floor((SOC_MAX - initial) / per + _SOC_TOL)
    rec["ev"] = ev
    rec["start"] = np.broadcast_to(np.asarray(start), (n,))
    rec["end"] = np.broadcast_to(np.asarray(end), (n,))
    rec["nmin"] = np.where(ev, np.clip(nmin, 0, 2**30), 0)
    rec["nmax"] = np.where(ev, np.clip(nmax, -1, 2**30), 0)
    rec["rating"] = np.where(ev, rating, 0.0)
    rec["capacity"] = np.where(ev, capacity, 1.0)
    rec["initial"] = np.where(ev, initial, 0.0)
    return rec


def voltage_limits(vset, vlow, vhigh):
    """lpsolver.py:185-186."""
    return vlow * vlow - vset * vset, vhigh * vhigh - vset * vset


@dataclass
class OperatorOptions:
    eps: float = 1e-8            # OSQP-style abs = rel tolerance on the scaled residuals
    max_iter: int = 20000
    check_every: int = 25
    adapt_every: int = 100
    alpha: float = 1.6
    rho_b_scale: float = 1.0     # rho_b = scale * kappa
    rho_v_scale: float = 25.0    # rho_v = scale * kappa / smax^2
    warm_start: bool = True
    use_graph: bool = True       # replay a hipGraph of `check_every` inner iterations (1 GPU)
    # The best rho depends strongly on how hard the voltage rows bind (x100 between the
    # 121144 feeder and a mildly stressed one), and re-tuning on the fly costs iterations.
    # So the first non-trivial solve tries each (rho_v, rho_b) scale below for two blocks
    # from the same state and keeps the one with the smallest residual.
    calibrate: bool = True
    cal_rho_v: tuple = (0.1, 1.0, 10.0, 100.0)
    cal_rho_b: tuple = (0.1, 1.0)
    cal_iters: int = 50
    # Node-space fast path: solve the QP over the M nodes only (2 T-column products per
    # iteration, no home traffic, no per-iteration collective) and accept the answer iff
    # no residence would be pushed below zero; otherwise fall back to the general path
    # for the rest of the run.
    node_fast: bool = True
    # "newton": semismooth Newton on the dual (voltage-row multipliers), the default;
    # "admm": the OSQP-form iterations above only.  The Newton path hands an iteration to
    # the ADMM forms when it cannot finish (more than 128 binding rows in a slot, ...).
    solver: str = "newton"
    newton_max: int = 60         # Newton iterations per operator solve
    newton_kadd: int = 6         # violated rows admitted to a slot's model per iteration
    chain: bool = True           # binding steady state: one Newton iteration enqueued unread
    newton_delta: float = 1e-10  # relative diagonal shift of the model Hessian
    newton_pivots: int = 300     # block-pivoting limit per model problem
    newton_ls: int = 30          # Armijo halvings
    # After an operator solve that needed no Newton iteration, the next ADMM iteration
    # launches the home sweep right behind the operator's first evaluation, before the host
    # has seen that evaluation's verdict (P_sch / G go to spare buffers): the GPU never
    # waits for the host.  If rows turn out to need work, the sweep is simply run again.
    speculate: bool = True
    # One GPU, multipliers all zero: the speculative sweep also does the home pass of the NEXT
    # operator evaluation (one pass over the homes per ADMM iteration instead of two).
    fuse_home_pass: bool = True
    # How the steady state judges the voltage rows of an estimate: "dense" = the f64 matrix-core
    # product R p (always possible); "tree" = two tree passes over the radial feeder, O(nodes)
    # instead of O(nodes^2), inside the sweep's own launch (needs the feeder: `feeder=` of
    # AdmmEngine); "auto" = tree when a feeder of at most REVS_TREE_MAX nodes was given.
    voltage: str = "auto"
    # streaming steady state: launches enqueued per native call.  Starts at stream_burst, x4 after
    # every call whose launches were all kept (up to stream_burst_max), back to stream_burst after
    # a failed verdict: the launches behind a failure are silenced on the device but still cost
    # ~3 us each, so a regime that fails often keeps its bursts short
    stream_burst: int = 8
    stream_burst_max: int = 512


def _dev_check(device):
    if not torch.cuda.is_available():
        raise _lib.RevsError("revs_admm_amd needs a ROCm GPU (torch.cuda.is_available() is "
                             "False); there is no CPU fallback")
    return torch.device(device)


def feeder_tree(parent, edge_r, cons_of, checked):
    """Host-side preparation of revs_tree_t: the feeder's nodes in DFS preorder.

    parent[i]   parent of tree node i, -1 when i hangs off the substation (a forest is fine)
    edge_r[i]   resistance of the edge from i to its parent
    cons_of[i]  constraint row (0..M-1) of node i, or -1
    checked[r]  whether row r is constrained (it carries residences, lpsolver.py:188-189)
    Returns dict(n, src, end, eo, cle, w) of numpy arrays (see include/revs_admm.h)."""
    parent = np.asarray(parent, np.int64)
    pad = (-len(parent)) % 8                 # the kernel's threads own 8 consecutive positions:
    if pad:                                  # pad with weightless nodes hanging off the substation
        parent = np.concatenate([parent, np.full(pad, -1, np.int64)])
        edge_r = np.concatenate([np.asarray(edge_r, np.float64), np.zeros(pad)])
        cons_of = np.concatenate([np.asarray(cons_of, np.int64), np.full(pad, -1, np.int64)])
    n = len(parent)
    kids = [[] for _ in range(n)]
    roots = []
    for i in range(n):
        (roots if parent[i] < 0 else kids[parent[i]]).append(i)
    order, size = [], np.ones(n, np.int64)
    stack = [(r, False) for r in reversed(roots)]
    while stack:
        u, done = stack.pop()
        if done:
            for c in kids[u]:
                size[u] += size[c]
            continue
        order.append(u)
        stack.append((u, True))
        stack.extend((c, False) for c in reversed(kids[u]))
    if len(order) != n:
        raise ValueError("feeder: parent[] does not describe a forest")
    order = np.asarray(order, np.int64)
    pos = np.empty(n, np.int64)
    pos[order] = np.arange(n)
    end = (pos + size)[order]                                  # by preorder position
    cons = np.asarray(cons_of, np.int64)[order]
    chk = np.asarray(checked, bool)
    src = np.where((cons >= 0) & chk[np.maximum(cons, 0)], cons, -1)
    eo = np.argsort(end, kind="stable")
    cle = np.searchsorted(end[eo], np.arange(n), side="right")
    w = 2.0 * np.asarray(edge_r, np.float64)[order]
    return dict(n=n, src=src.astype(np.int32), end=end.astype(np.int32), eo=eo.astype(np.int32),
                cle=cle.astype(np.int32), w=w)


def tree_voltage_host(tree, p):
    """numpy restatement of the three prefix sums (tests, and the constructor's check that the
    feeder reproduces Rn): v at the checked rows, indexed like p."""
    n, src = tree["n"], tree["src"]
    inj = np.where(src >= 0, 1.0, 0.0)[:, None] * p[np.maximum(src, 0)]
    C = np.concatenate([np.zeros((1, p.shape[1])), np.cumsum(inj, 0)])
    wp = tree["w"][:, None] * (C[tree["end"]] - C[:-1])
    pre = np.cumsum(wp, 0)
    F = np.concatenate([np.zeros((1, p.shape[1])), np.cumsum(wp[tree["eo"]], 0)])
    v = pre - F[tree["cle"]]
    out = np.zeros_like(p)
    out[src[src >= 0]] = v[src >= 0]
    return out


class AdmmEngine:
    """State of one ADMM run on one GPU.

    Parameters
    ----------
    cost      (T,)   tariff
    homes     (n,)   HOME_DTYPE records (pack_homes) -- this rank's residences
    load      (n,T)  base load
    node_of   (n,)   constraint-node index of each residence (0..M-1)
    Rn        (M,M)  LinDistFlow matrix restricted to the constraint nodes
    group            torch.distributed process group when residences are sharded
    feeder           optional (parent, edge_r, cons_of): the radial feeder behind Rn as a tree --
                     parent[i] (-1: hangs off the substation), resistance of the edge to the
                     parent, constraint row of tree node i (or -1).  With it the steady state
                     evaluates R p in O(nodes) (OperatorOptions.voltage); the constructor checks
                     that the tree reproduces Rn.
    """

    def __init__(self, cost, homes, load, node_of, Rn, kappa=5.0, vset=1.0, vlow=0.95,
                 vhigh=1.05, mode="binary", device="cuda:0", pdhg=None,
                 op: OperatorOptions | None = None, group=None, node_counts=None,
                 pdhg_warm=True, feeder=None, _kernels=None):
        if _kernels is None:
            self.lib = _lib.load()               # raises when the HIP library is missing
            self.dev = _dev_check(device)        # raises without a GPU
            torch.cuda.set_device(self.dev)
        else:
            # tests only (tests/fake_kernels.py): an object exposing the C ABI's entry
            # points over host memory, so the driver logic -- sharding, all-reduce,
            # stopping rules -- can run under gloo without a GPU.  Never set by the product.
            self.lib = _kernels
            self.dev = torch.device(device)
        self.group = group
        self.kappa = float(kappa)
        self.mode = MODES[mode] if isinstance(mode, str) else int(mode)
        self.op = op or OperatorOptions()
        if self.op.solver == "newton" and np.asarray(Rn).shape[0] > 16384:
            self.op = dataclasses.replace(self.op, solver="admm")   # revs_op_dual_select's limit
        self.vlo, self.vhi = voltage_limits(vset, vlow, vhigh)

        load = np.ascontiguousarray(load, np.float32)
        n, T = load.shape
        self.n, self.T = n, T
        node_of = np.asarray(node_of, np.int64)
        Rn = np.asarray(Rn, np.float64)
        M = Rn.shape[0]
        self.M = M
        assert homes.dtype == HOME_DTYPE and len(homes) == n and len(node_of) == n
        assert node_of.min(initial=0) >= 0 and node_of.max(initial=0) < M

        # residences sorted by node -> CSR; remember the permutation.  Inside a node the
        # residences with an EV come first, then the others (whose home problem is trivial:
        # no PDHG pass): a wavefront of the sweep carries 8 consecutive homes and iterates until
        # the slowest has converged, so like-with-like keeps whole wavefronts out of the loop.
        # (EV homes first on even nodes, last on odd ones: the like runs of neighbouring nodes
        # join up, which halves the number of mixed wavefronts)
        self.perm = np.lexsort((np.arange(n), (homes["ev"] == 0) ^ ((node_of & 1) == 1), node_of))
        self.inv_perm = np.empty_like(self.perm)
        self.inv_perm[self.perm] = np.arange(n)
        local_counts = np.bincount(node_of, minlength=M).astype(np.int64)
        node_ptr = np.concatenate([[0], np.cumsum(local_counts)]).astype(np.int64)
        counts = local_counts.copy()
        if group is not None:
            ct = torch.from_numpy(counts).to(self.dev)
            torch.distributed.all_reduce(ct, group=group)
            counts = ct.cpu().numpy()
        if node_counts is not None:
            counts = np.asarray(node_counts, np.int64)
        self.node_counts = counts

        f32 = dict(dtype=torch.float32, device=self.dev)
        f64 = dict(dtype=torch.float64, device=self.dev)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.cost = up(np.asarray(cost, np.float32))
        self.homes = up(homes[self.perm].view(np.uint8).reshape(n, HOME_DTYPE.itemsize))
        self.load = up(load[self.perm])
        self.node_ptr = up(node_ptr)
        self.P_est = torch.zeros(n, T, **f32)          # lpsolver.py:244
        self.P_est_new = torch.zeros(n, T, **f32)
        self.P_sch = torch.zeros(n, T, **f32)          # lpsolver.py:245
        self.G = torch.zeros(n, T, **f32)              # lpsolver.py:246
        self.S = torch.zeros(n, T, **f32)
        self.Csoc = torch.zeros(n, T + 1, **f32)
        self.diff = torch.zeros(n, **f32)
        self.status = torch.zeros(n, dtype=torch.int32, device=self.dev)
        self.dsq = torch.zeros(n, **f32)
        self.res_scratch = torch.zeros(3 * int(self.lib.revs_residual_num_chunks(n)),
                                       dtype=torch.float64, device=self.dev)
        self.resid = torch.zeros(4, **f32)
        # PDHG multipliers carried across ADMM iterations (warm start), relaxed PDHG only
        self._pdhg_warm = bool(pdhg_warm) and self.mode == _lib.MODE_RELAXED_PDHG
        self.pdhg_dual = None
        self.pdhg = PDHG()
        self.lib.revs_pdhg_defaults(C.byref(self.pdhg))
        if pdhg:
            for k, v in pdhg.items():
                setattr(self.pdhg, k, v)
        if self._pdhg_warm:      # one scalar per home, or one per SOC row with full_rows
            self.pdhg_dual = torch.zeros((n, T) if self.pdhg.full_rows else (n,), **f32)

        # ---- operator setup ----
        self._Rn_host, self._counts_host = Rn, counts
        self._admm_ready = False
        nz = lambda: torch.zeros(M, T, **f64)
        self.ksplit1 = int(min(8, max(1, -(-256 // ((M + 31) // 32 if T <= 32 else (M + 15) // 16)))))
        nz1 = lambda: torch.zeros(self.ksplit1, M, T, **f64)
        self._fast_ok = bool(self.op.node_fast)
        self._fast_cold = True
        self.op_cold = True
        self.op_path_hist: list[str] = []
        self.op_iters_hist: list[int] = []
        self.iteration = 0
        if self.op.solver != "newton":
            self._ensure_admm()
        # dual Newton path: R itself (d = R^T y; candidate rows of K), and R^T with the rows
        # of nodes without residences zeroed (v = R p is constrained where residences are)
        has = (counts > 0).astype(np.float64)
        self.R64 = up(Rn)
        self.R64T = up((Rn * has[:, None]).T)
        A = _lib.DUAL_AMAX
        self.yd = [nz(), nz()]                         # multipliers: current, trial
        self.pnq = torch.zeros(3, M, T, **f64)
        self.d_sl, self.v_sl = nz1(), nz1()
        self.vfull = nz()
        self.violw = nz()
        self.d_part = torch.zeros(int(self.lib.revs_op_dual_blocks(M)), T, 4, **f64)
        self.nks = int(min(16, max(1, M // 128)))       # column slabs of the model Hessian
        self.k_slabs = torch.zeros(T, self.nks, A, A, **f64)
        self.k_full = torch.zeros(T, A, A, **f64)
        self.tile_cnt = torch.zeros((M + 31) // 32, dtype=torch.int32, device=self.dev)
        self.c_idx = [torch.zeros(T, A, dtype=torch.int64, device=self.dev) for _ in range(2)]
        self.c_cnt = [torch.zeros(T, dtype=torch.int32, device=self.dev) for _ in range(2)]
        self.c_val = [torch.zeros(T, 3, A, **f64) for _ in range(2)]
        self.yhat = torch.zeros(T, A, **f64)
        self._y_support = False
        self._chain_ok, self._chain_few, self._pre_kept = False, False, False
        self._chain_seq = 0.0                          # tags of the chain's evaluations: -1, -2, ...
        self.chain_hist = [0, 0]                       # chained Newton iterations kept / redone
        self._spec_ok = False
        self._spec_wait, self._spec_back = 0, 1
        self._sup = None
        self.model_calls = [0, 0]                      # small / general model kernels used
        self.spec_hist = [0, 0]                        # speculative sweeps kept / discarded
        self.P_sch_alt = torch.zeros(n, T, **f32) if self.op.solver == "newton" else None
        self.G_alt = torch.zeros(n, T, **f32) if self.op.solver == "newton" else None
        cuda = self.dev.type == "cuda"
        # per-slot stats are written by the select kernel straight into pinned host memory
        self.stats_host = [torch.zeros(T, 8, dtype=torch.float64, pin_memory=cuda) for _ in range(2)]
        self.stats_dev = []
        for t in self.stats_host:
            if cuda:
                dp = C.c_void_p()
                check(self.lib.revs_host_device_ptr(t.data_ptr(), C.byref(dp)),
                      "revs_host_device_ptr")
                self.stats_dev.append(int(dp.value))
            else:
                self.stats_dev.append(t.data_ptr())
        self.stats_ev = [torch.cuda.Event() if cuda else None for _ in range(2)]
        # step lengths (host -> kernel) and pivot counts (kernel -> host) live in pinned host
        # memory as well: no copy launches inside a Newton iteration
        self.alpha_h = torch.zeros(T, dtype=torch.float64, pin_memory=cuda)
        self.info_h = torch.zeros(T, dtype=torch.int32, pin_memory=cuda)
        self.alpha_dev, self.info_dev = self.alpha_h.data_ptr(), self.info_h.data_ptr()
        if cuda:
            for name, t in (("alpha_dev", self.alpha_h), ("info_dev", self.info_h)):
                dp = C.c_void_p()
                check(self.lib.revs_host_device_ptr(t.data_ptr(), C.byref(dp)),
                      "revs_host_device_ptr")
                setattr(self, name, int(dp.value))
        self.newton_hist: list[tuple] = []
        if cuda and self.op.solver == "newton" and _kernels is None:
            # first use of a kernel loads its code object (1-2 ms each on this stack): touch the
            # Newton-iteration kernels now, with empty candidate lists, not inside the first solve
            check(self.lib.revs_op_dual_model(M, T, ptr(self.R64), ptr(self.pnq[1]), ptr(self.c_idx[0]),
                                              ptr(self.c_cnt[0]), ptr(self.c_val[0]), self.kappa,
                                              self.op.newton_delta, self.op.newton_pivots, self.nks,
                                              ptr(self.k_slabs), ptr(self.k_full), ptr(self.yhat),
                                              self.info_dev, self.stream), "revs_op_dual_model")
            check(self.lib.revs_op_dual_step(T, ptr(self.c_idx[0]), ptr(self.c_cnt[0]),
                                             ptr(self.c_val[0]), ptr(self.yhat), self.alpha_dev,
                                             ptr(self.yd[1]), self.stats_dev[1] + 32, self.stream),
                  "revs_op_dual_step")
            check(self.lib.revs_op_dual_model_small(M, T, ptr(self.R64), ptr(self.pnq[1]),
                                                    ptr(self.c_idx[0]), ptr(self.c_cnt[0]),
                                                    ptr(self.c_val[0]), self.kappa,
                                                    self.op.newton_delta, self.op.newton_pivots,
                                                    ptr(self.k_full), ptr(self.yhat), self.info_dev,
                                                    self.stream), "revs_op_dual_model_small")
            self._gemm1(self.R64, self.yd[0], self.d_sl)
            torch.cuda.synchronize(self.dev)
        # steady-state iteration as ONE native call (one GPU; see revs_plan_spec_step)
        self._plan = None
        self._fused_ready = False        # the last kept sweep did the next evaluation's home pass
        self._fused_p = None             # ... and where it left the node sums
        self.recompute_pe_new = False
        self._ar_ahead = False           # ... already exchanged between the ranks
        self._prod_ahead = False         # ... and the product on them already enqueued
        if (cuda and self.op.solver == "newton" and _kernels is None
                and not os.environ.get("REVS_NO_PLAN")):
            d = _lib.PlanDesc()
            d.n_homes, d.m, d.T = n, M, T
            d.node_ptr, d.R, d.Rt = ptr(self.node_ptr), ptr(self.R64), ptr(self.R64T)
            d.kappa, d.vlo, d.vhi = self.kappa, self.vlo, self.vhi
            d.kadd, d.ksplit = self.op.newton_kadd, self.ksplit1
            d.d_slabs, d.v_slabs, d.pnq = ptr(self.d_sl), ptr(self.v_sl), ptr(self.pnq)
            d.vfull, d.viol, d.partial = ptr(self.vfull), ptr(self.violw), ptr(self.d_part)
            d.cand_idx, d.cand_cnt, d.cand_val = (ptr(self.c_idx[0]), ptr(self.c_cnt[0]),
                                                  ptr(self.c_val[0]))
            d.stats, d.stats_host = self.stats_dev[0], self.stats_host[0].data_ptr()
            d.cost, d.homes, d.load = ptr(self.cost), ptr(self.homes), ptr(self.load)
            d.diff, d.dsq, d.status = ptr(self.diff), ptr(self.dsq), ptr(self.status)
            d.pdhg_dual, d.mode, d.pdhg = ptr(self.pdhg_dual), self.mode, self.pdhg
            self.node_of_dev = up(node_of[self.perm].astype(np.int32))
            self.P_est_alt = torch.zeros(n, T, **f32)
            self.p_alt = nz()                         # second buffer of the fused node sums
            d.node_of = ptr(self.node_of_dev)
            # with no multipliers the sweep recomputes the operator's answer instead of reading it
            # (same bits, one input stream less: 20.0 -> 19.0 us per launch at 100k homes x 24,
            # round 2; REVS_RECOMPUTE=0 restores the read)
            rec = os.environ.get("REVS_RECOMPUTE", "").strip()
            d.recompute_pe_new = int(rec not in ("0", "false", "no")) if rec else 1
            self.recompute_pe_new = bool(d.recompute_pe_new)
            d.cand_idx1, d.cand_cnt1, d.cand_val1 = (ptr(self.c_idx[1]), ptr(self.c_cnt[1]),
                                                     ptr(self.c_val[1]))
            d.stats1, d.stats1_host = self.stats_dev[1], self.stats_host[1].data_ptr()
            d.yhat, d.k_full, d.info = ptr(self.yhat), ptr(self.k_full), self.info_dev
            d.delta, d.eps, d.max_pivots = self.op.newton_delta, self.op.eps, self.op.newton_pivots
            self._plan_desc = d
            self._plan = self.lib.revs_plan_create(C.byref(d))
            if not self._plan:
                raise _lib.RevsError("revs_plan_create failed: "
                                     + self.lib.revs_last_error().decode())
        # third node-sum buffer and the feeder as a tree: streaming steady state
        self.p_alt2 = nz()
        self._burst = max(1, int(self.op.stream_burst))
        self._p_clear = None             # the node-sum array the last streaming launch cleared
        self._tree = None
        self._comm = None
        if feeder is not None and self.op.voltage in ("auto", "tree"):
            par, er, cons = feeder
            if len(par) <= _lib.TREE_MAX:          # (padded to a multiple of 8 below)
                tr = feeder_tree(par, er, cons, counts > 0)
                probe = np.random.default_rng(0).uniform(0.5, 1.5, (M, 2)) * (counts > 0)[:, None]
                ref = (Rn @ probe) * (counts > 0)[:, None]
                got = tree_voltage_host(tr, probe)
                if np.abs(got - ref).max() > 1e-9 * max(np.abs(ref).max(), 1e-300):
                    raise ValueError("feeder: the tree does not reproduce Rn (R[i][j] = 2 x the "
                                     "resistance shared by the substation->i and ->j paths)")
                self._tree_host = tr
                self._tree_dev = {k: up(v) for k, v in tr.items() if k != "n"}
                self._tree = _lib.Tree(tr["n"], *[ptr(self._tree_dev[k]) for k in ("src", "end", "eo", "cle", "w")])
                if self._plan is not None:
                    check(self.lib.revs_plan_set_tree(self._plan, C.byref(self._tree)),
                          "revs_plan_set_tree")
            elif self.op.voltage == "tree":
                raise ValueError(f"feeder has {len(par)} nodes; the tree form holds {_lib.TREE_MAX}")
        elif self.op.voltage == "tree":
            raise ValueError('OperatorOptions(voltage="tree") needs feeder=')
        if group is not None and cuda and _kernels is None and not os.environ.get("REVS_NO_COMM"):
            # the library's own RCCL communicator: unique id from rank 0 over the caller's group
            ws, rk = torch.distributed.get_world_size(group), torch.distributed.get_rank(group)
            idb = (C.c_char * 128)()
            if rk == 0:
                check(self.lib.revs_comm_unique_id(idb), "revs_comm_unique_id")
            idt = torch.frombuffer(bytearray(idb.raw), dtype=torch.uint8).to(self.dev)
            torch.distributed.broadcast(idt, src=torch.distributed.get_global_rank(group, 0), group=group)
            raw = bytes(idt.cpu().numpy().tobytes())
            self._comm = self.lib.revs_comm_create(raw, rk, ws)
            if not self._comm:
                raise _lib.RevsError("revs_comm_create failed: " + self.lib.revs_last_error().decode())
            if self._plan is not None:
                check(self.lib.revs_plan_set_comm(self._plan, self._comm), "revs_plan_set_comm")
        # R (float) for the voltage check
        self.R32 = up(Rn.astype(np.float32))
        self.node_load = torch.zeros(M, T, **f32)
        self.volt = torch.zeros(M, T, **f32)

    def _ensure_admm(self):
        """State of the ADMM forms (eigendecomposition of the scaled R on the host, per-home
        double arrays): built on first use -- the default dual Newton path never needs it."""
        if self._admm_ready:
            return
        self._admm_ready = True
        Rn, counts = self._Rn_host, self._counts_host
        n, T, M = self.n, self.T, self.M
        f64 = dict(dtype=torch.float64, device=self.dev)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        # Voltage row m is scaled by sqrt(n_m) (bounds too), so the operator matrix
        # D^1/2 R D^1/2 is symmetric PSD = Q L Q^T: one factor serves C_v and C_v^T.
        # Nodes without residences get a zero row: voltage is constrained where
        # residences are, as in the reference (R_res, lpsolver.py:188-189).
        sq = np.sqrt(counts.astype(np.float64))
        lam, Q = np.linalg.eigh(sq[:, None] * Rn * sq[None, :])
        lam = np.maximum(lam, 0.0)
        self.smax = float(lam.max())
        self.Q, self.QT = up(Q), up(Q.T)
        # the scaled sensitivity matrix itself, for the one-product voltage check Rs.p0
        self.Rs = up((Q * lam[None, :]) @ Q.T)
        self.s = up(lam)
        self.sqrt_n = up(sq)
        self.inv_sqrt_n = up(np.where(counts > 0, 1.0 / np.maximum(sq, 1e-300), 0.0))
        self.g0 = torch.zeros(n, T, **f64)
        self.sb = torch.zeros(n, T, **f64)             # z_b + y_b of the g >= 0 rows
        nz = lambda: torch.zeros(M, T, **f64)
        (self.zv, self.yv, self.cx, self.w, self.rhat, self.xc, self.a, self.sa,
         self.cty) = (nz() for _ in range(9))
        # GEMM outputs come as K-split partial slabs (summed by the node kernels): enough
        # workgroups to fill 256 CUs even when M/32 row tiles x 2 products is below that
        self.cat = 2 * T <= 192                        # [rhat | w] in one product
        ncol = 2 * T if self.cat else T
        tiles = ((M + 31) // 32 if ncol <= 48 else (M + 15) // 16) * (1 if self.cat else 2)
        self.ksplit = int(min(8, max(1, -(-256 // tiles))))
        nzs = lambda: torch.zeros(self.ksplit, M, T, **f64)
        self.ta, self.tb, self.va, self.usa = nzs(), nzs(), nzs(), nzs()
        # node-space fast path
        nz1 = lambda: torch.zeros(self.ksplit1, M, T, **f64)
        self.f_wh, self.f_zt = nz1(), nz1()
        (self.p0, self.gmin, self.ph0, self.xh, self.sx, self.dnode, self.slack) = (nz() for _ in range(7))
        self.f_stats = torch.zeros(2, **f64)
        self.rho_f = torch.full((T,), self.op.rho_v_scale * self.kappa / self.smax ** 2, **f64)
        self._dnode_zero = False
        # exact presolve of the fast path (see revs_op_node_prep): needs R >= 0 and vlo <= 0
        self.preclamp = int(bool(Rn.min() >= 0.0 and self.vlo <= 0.0))
        self._fast_wait = 0
        self._fast_backoff = 1
        self._fast_cal = False
        self._fgraph = None
        self._fgraph_warm = False
        self.rho_v = torch.full((T,), self.op.rho_v_scale * self.kappa / self.smax ** 2, **f64)
        self.rho_b = torch.full((T,), self.op.rho_b_scale * self.kappa, **f64)
        self.res_out = torch.zeros(8, T, **f64)
        self._calibrated = False
        self._graph = None
        self._graph_warm = False

    # ------------------------------------------------------------------ util
    @property
    def stream(self):
        if self.dev.type != "cuda":
            return None
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _gemm(self, At, B, Cout, accumulate=0):
        k, m = At.shape
        n = B.shape[1]
        check(self.lib.revs_gemm_tn_f64(m, n, k, ptr(At), m, ptr(B), n, ptr(Cout), n,
                                        accumulate, self.stream), "revs_gemm_tn_f64")

    def _gemm_cat(self, At, B0, B1, C0, C1):
        """[C0 | C1] = At^T [B0 | B1] as K-split slabs; one launch when 2T <= 192."""
        M, T, st = self.M, self.T, self.stream
        if self.cat:
            rc = self.lib.revs_gemm_tn_f64_cat(M, T, M, ptr(At), ptr(B0), ptr(B1), ptr(C0),
                                               ptr(C1), self.ksplit, st)
            check(rc, "revs_gemm_tn_f64_cat")
        else:
            rc = self.lib.revs_gemm_tn_f64_x2(M, T, M, ptr(At), ptr(B0), ptr(C0), ptr(At),
                                              ptr(B1), ptr(C1), self.ksplit, st)
            check(rc, "revs_gemm_tn_f64_x2")

    def _allreduce(self, t, op=None):
        if self.group is None:
            return
        RO = torch.distributed.ReduceOp
        if (getattr(self, "_comm", None) and t.dtype == torch.float64 and t.is_contiguous()
                and op in (None, RO.SUM, RO.MAX, RO.MIN)):
            # the library's own RCCL communicator, on the compute stream: no second stream, no
            # event hand-over (torch.distributed keeps only the bootstrap of the unique id)
            code = 0 if op in (None, RO.SUM) else (2 if op == RO.MAX else 3)
            check(self.lib.revs_comm_allreduce_f64(self._comm, ptr(t), t.numel(), code, self.stream),
                  "revs_comm_allreduce_f64")
            return
        torch.distributed.all_reduce(t, op=op or RO.SUM, group=self.group)

    # -------------------------------------------------------------- operator
    def _home_pass(self, with_update: bool, check: bool = False, reduce: bool = True):
        o = self.op
        rc = self.lib.revs_op_home_pass(
            self.M, self.T, ptr(self.node_ptr), ptr(self.inv_sqrt_n), ptr(self.sb), ptr(self.g0),
            ptr(self.xc) if with_update else None, ptr(self.rho_b), self.kappa, o.alpha,
            ptr(self.rhat), ptr(self.cty) if check else None,
            ptr(self.res_out) if check else None, self.stream)
        _lib.check(rc, "revs_op_home_pass")
        if reduce:
            self._allreduce(self.rhat)

    def _home_pass_fused(self, reduce: bool = True):
        o = self.op
        rc = self.lib.revs_op_home_pass_fused(
            self.M, self.T, ptr(self.node_ptr), ptr(self.inv_sqrt_n), ptr(self.sb), ptr(self.g0),
            ptr(self.rho_b), self.kappa, o.alpha, ptr(self.rhat), self.ksplit, ptr(self.va),
            ptr(self.usa), ptr(self.rho_v), ptr(self.sqrt_n), self.vlo, self.vhi, ptr(self.xc),
            ptr(self.zv), ptr(self.yv), ptr(self.w), self.stream)
        _lib.check(rc, "revs_op_home_pass_fused")
        if reduce:
            self._allreduce(self.rhat)

    def _node_half(self, check: bool = False, fuse: bool = False):
        """rhat -> xc, and the z_v / y_v update (GEMMs on the f64 matrix cores).  On a
        checking iteration also the node-side residual maxima and cty = C_v^T y_v.  With
        `fuse` the node update is left to the following fused home pass."""
        o, lib, M, T, st = self.op, self.lib, self.M, self.T, self.stream
        ks = self.ksplit
        self._gemm_cat(self.Q, self.rhat, self.w, self.ta, self.tb)        # Q^T [rhat | w]
        _lib.check(lib.revs_op_node_scale(M, T, ks, ptr(self.ta), ptr(self.tb), ptr(self.s),
                                          ptr(self.rho_v), ptr(self.rho_b), self.kappa,
                                          ptr(self.a), ptr(self.sa), st), "revs_op_node_scale")
        self._gemm_cat(self.QT, self.a, self.sa, self.va, self.usa)        # Q [a | l a]
        if fuse:
            return
        _lib.check(lib.revs_op_node_update(M, T, ks, ptr(self.va), ptr(self.rhat), ptr(self.usa),
                                           ptr(self.rho_v), ptr(self.rho_b), ptr(self.sqrt_n),
                                           self.kappa, o.alpha, self.vlo, self.vhi, ptr(self.xc),
                                           ptr(self.zv), ptr(self.yv), ptr(self.w),
                                           ptr(self.res_out) if check else None, st),
                   "revs_op_node_update")
        if check:
            self._gemm(self.Q, self.yv, self.tb[0])                 # Q^T y_v
            _lib.check(lib.revs_op_row_scale(M, T, ptr(self.s), ptr(self.tb[0]),
                                             ptr(self.ta[0]), st), "revs_op_row_scale")
            self._gemm(self.QT, self.ta[0], self.cty)               # Q L Q^T y_v = C_v^T y_v

    def _inner_block(self):
        """`check_every` inner iterations, the last one also accumulating the residual
        maxima.  hipGraphs (through torch.cuda.CUDAGraph: the ctypes launches go to torch's
        current stream, which is the capture stream) cut the host work:
          * one GPU: the whole block is one graph -- one host call per 25 iterations;
          * sharded: one iteration's kernels (2 products, node scale, fused home pass) are
            a graph and only the RCCL all-reduce of rhat between iterations stays eager:
            2 host calls per iteration, and no collective is ever captured."""
        n_it = self.op.check_every

        def body():
            self.res_out.zero_()
            for k in range(n_it):
                if k == n_it - 1:              # checking iteration: separate passes + residuals
                    self._node_half(check=True)
                    self._home_pass(with_update=True, check=True)
                else:                          # node update fused into the home pass
                    self._node_half(fuse=True)
                    self._home_pass_fused()

        if not self.op.use_graph or self.dev.type != "cuda":
            return body()
        if not self._graph_warm:               # first block eager: warms up, loads code objects
            self._graph_warm = True
            return body()
        if self.group is None:
            if self._graph is None:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    body()
                self._graph = g
            return self._graph.replay()
        if self._graph is None:
            gs = []
            for chk in (False, True):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._node_half(check=chk, fuse=not chk)
                    if chk:
                        self._home_pass(with_update=True, check=True, reduce=False)
                    else:
                        self._home_pass_fused(reduce=False)
                gs.append(g)
            self._graph = gs
        self.res_out.zero_()
        for k in range(n_it):
            self._graph[1 if k == n_it - 1 else 0].replay()
            self._allreduce(self.rhat)

    def _residuals(self):
        self._allreduce(self.res_out, torch.distributed.ReduceOp.MAX if self.group else None)
        return self.res_out.cpu().numpy()

    def _rel_residuals(self, r):
        vscale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        n_pv = np.maximum(np.maximum(r[3], r[4]), vscale)
        n_pb = np.maximum(r[5], 1e-12)
        n_d = np.maximum(np.maximum(self.kappa * r[5], r[6]), np.maximum(r[7], 1e-12))
        return np.maximum(r[0] / n_pv, r[1] / n_pb), r[2] / n_d

    def _set_rho(self, rv_scale, rb_scale):
        self.rho_v.fill_(rv_scale * self.kappa / self.smax ** 2)
        self.rho_b.fill_(rb_scale * self.kappa)
        _lib.check(self.lib.revs_op_node_w(self.M, self.T, ptr(self.zv), ptr(self.yv),
                                           ptr(self.rho_v), ptr(self.w), self.stream),
                   "revs_op_node_w")
        self._home_pass(with_update=False)

    def _calibrate_rho(self):
        """Try each candidate (rho_v, rho_b) for two blocks from the current state, keep the
        best.  Returns the number of inner iterations spent."""
        o = self.op
        snap = [t.clone() for t in (self.sb, self.zv, self.yv)]
        best, spent = None, 0
        for rv in o.cal_rho_v:
            for rb in o.cal_rho_b:
                for t, c in zip((self.sb, self.zv, self.yv), snap):
                    t.copy_(c)
                self._set_rho(rv, rb)
                nblk = max(1, -(-o.cal_iters // o.check_every))
                for _ in range(nblk):
                    self._inner_block()
                spent += nblk * o.check_every
                rel_p, rel_d = self._rel_residuals(self._residuals())
                score = float(max(rel_p.max(), rel_d.max()))
                if np.isfinite(score) and (best is None or score < best[0]):
                    best = (score, rv, rb)
        for t, c in zip((self.sb, self.zv, self.yv), snap):
            t.copy_(c)
        self.rho_scales = best[1:]
        self.cal_score = best[0]
        self._set_rho(*self.rho_scales)
        self._calibrated = True
        return spent

    # ------------------------------------------------- operator, node-space fast path
    def _gemm1(self, At, B, Cslabs):
        check(self.lib.revs_gemm_tn_f64_split(self.M, self.T, self.M, ptr(At), ptr(B),
                                              ptr(Cslabs), self.ksplit1, self.stream),
              "revs_gemm_tn_f64_split")

    def _fast_iteration(self, chk: bool):
        o, lib, M, T, st, ks = self.op, self.lib, self.M, self.T, self.stream, self.ksplit1
        self._gemm1(self.Q, self.w, self.f_wh)                             # wh = Q^T w
        check(lib.revs_op_nodefast_scale(M, T, ks, ptr(self.f_wh), ptr(self.ph0), ptr(self.s),
                                         ptr(self.rho_f), self.kappa, ptr(self.xh), ptr(self.sx),
                                         st), "revs_op_nodefast_scale")
        self._gemm1(self.QT, self.sx, self.f_zt)                           # zt = Q (l xh)
        check(lib.revs_op_nodefast_update(M, T, ks, ptr(self.f_zt), ptr(self.rho_f),
                                          ptr(self.sqrt_n), o.alpha, self.vlo, self.vhi,
                                          ptr(self.zv), ptr(self.yv), ptr(self.w),
                                          ptr(self.res_out) if chk else None, st),
              "revs_op_nodefast_update")
        if chk:
            self._gemm1(self.Q, self.yv, self.f_wh)                        # yh = Q^T y_v
            check(lib.revs_op_nodefast_dualres(M, T, ks, ptr(self.xh), ptr(self.ph0),
                                               ptr(self.s), ptr(self.f_wh), self.kappa,
                                               ptr(self.res_out), st), "revs_op_nodefast_dualres")

    def _fast_block(self):
        n_it = self.op.check_every

        def body():
            self.res_out.zero_()
            for k in range(n_it):
                self._fast_iteration(k == n_it - 1)

        if not self.op.use_graph or self.dev.type != "cuda":
            return body()
        if not self._fgraph_warm:
            self._fgraph_warm = True
            return body()
        if self._fgraph is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            self._fgraph = g
        self._fgraph.replay()

    def _fast_set_rho(self, scale):
        self.rho_f.fill_(scale * self.kappa / self.smax ** 2)
        check(self.lib.revs_op_node_w(self.M, self.T, ptr(self.zv), ptr(self.yv), ptr(self.rho_f),
                                      ptr(self.w), self.stream), "revs_op_node_w")

    def _fast_residuals(self):
        r = self.res_out.cpu().numpy()          # identical on every rank: no reduction needed
        vscale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        rel_p = r[0] / np.maximum(np.maximum(r[3], r[4]), vscale)
        rel_d = r[2] / np.maximum(np.maximum(self.kappa * r[5], r[6]), np.maximum(r[7], 1e-12))
        return rel_p, rel_d

    def _operator_solve_node(self, precheck=False):
        """Fast path.  Returns True when its answer (in P_est_new) is the operator's exact
        answer; "pre" when the pre-check found a residence with g0 < 0 (nothing solved yet),
        "post" when the node solve finished but some residence would have to be clamped."""
        o, lib, M, T, st = self.op, self.lib, self.M, self.T, self.stream
        check(lib.revs_op_node_prep(M, T, ptr(self.node_ptr), ptr(self.inv_sqrt_n), ptr(self.P_est),
                                    ptr(self.P_sch), ptr(self.G), self.kappa, self.preclamp,
                                    ptr(self.p0), ptr(self.gmin), None, st), "revs_op_node_prep")
        if self.group is not None:               # the only exchange of this outer iteration
            self._allreduce(self.p0)
            if not self.preclamp:                # with the pre-clamp gmin >= 0 is known, and its
                self._allreduce(self.gmin, torch.distributed.ReduceOp.MIN)   # exact value is
                                                 # only needed if rows bind (below)
        # The voltage check proper: v0 = Rs.p0, one product on the f64 matrix cores.  If it
        # already respects every row, the projection is g0 itself (d = 0): no iteration.
        self._gemm1(self.Rs, self.p0, self.f_zt)                           # Rs symmetric: At = Rs
        self.f_stats.zero_()
        check(lib.revs_op_nodefast_feas(M, T, self.ksplit1, ptr(self.f_zt), ptr(self.sqrt_n),
                                        ptr(self.gmin), self.vlo, self.vhi, ptr(self.cx),
                                        ptr(self.f_stats), st), "revs_op_nodefast_feas")
        viol0, neg0 = self.f_stats.cpu().tolist()        # the one host sync of the easy case
        if neg0 > 0.0 and (precheck or viol0 == 0.0):
            return "pre"                     # a residence with g0 < 0 has to be clamped anyway
        if viol0 == 0.0:
            if not self._dnode_zero:
                self.dnode.zero_()
                self._dnode_zero = True
            check(lib.revs_op_node_apply(M, T, ptr(self.node_ptr), ptr(self.inv_sqrt_n),
                                         ptr(self.P_est), ptr(self.P_sch), ptr(self.G), self.kappa,
                                         self.preclamp, ptr(self.dnode), ptr(self.P_est_new), st),
                  "revs_op_node_apply")
            self._fast_cold = True           # z = Rs p0, y = 0 is re-made when rows bind again
            self.op_iters_hist.append(0)
            self.op_path_hist.append("node")
            self.op_converged = True
            return True
        if self.group is not None and self.preclamp:
            self._allreduce(self.gmin, torch.distributed.ReduceOp.MIN)   # for the slack test
        self._dnode_zero = False
        self._gemm(self.Q, self.p0, self.ph0)                              # ph0 = Q^T p0
        if self._fast_cold:
            check(lib.revs_op_init_node(M, T, ptr(self.cx), ptr(self.rho_f), ptr(self.sqrt_n),
                                        self.vlo, self.vhi, ptr(self.zv), ptr(self.yv),
                                        ptr(self.w), st), "revs_op_init_node")   # z = clip(Rs p0), y = 0
            self._fast_cold = False
        it, converged = 0, False
        while it < o.max_iter:
            self._fast_block()
            it += o.check_every
            rel_p, rel_d = self._fast_residuals()
            if max(rel_p.max(), rel_d.max()) <= o.eps:
                converged = True
                break
            if o.calibrate and not self._fast_cal:
                snap = [t.clone() for t in (self.zv, self.yv)]
                best = None
                nblk = max(1, -(-o.cal_iters // o.check_every))
                for rv in o.cal_rho_v:
                    for t, c in zip((self.zv, self.yv), snap):
                        t.copy_(c)
                    self._fast_set_rho(rv)
                    for _ in range(nblk):
                        self._fast_block()
                    it += nblk * o.check_every
                    rp, rd = self._fast_residuals()
                    score = float(max(rp.max(), rd.max()))
                    if np.isfinite(score) and (best is None or score < best[0]):
                        best = (score, rv)
                for t, c in zip((self.zv, self.yv), snap):
                    t.copy_(c)
                self._fast_set_rho(best[1])
                self.rho_f_scale = best[1]
                self._fast_cal = True
                continue
            if o.adapt_every and it % o.adapt_every == 0:
                sc = np.sqrt(np.maximum(rel_p, 1e-14) / np.maximum(rel_d, 1e-14))
                sc = np.clip(sc, 0.2, 5.0)
                sc = np.where((sc > 2.0) | (sc < 0.5), sc, 1.0)
                if (sc != 1.0).any():
                    self.rho_f.mul_(torch.from_numpy(sc).to(self.dev))
                    check(lib.revs_op_node_w(M, T, ptr(self.zv), ptr(self.yv), ptr(self.rho_f),
                                             ptr(self.w), st), "revs_op_node_w")
        self._gemm1(self.QT, self.xh, self.f_zt)                           # x = Q xh
        self.f_stats.zero_()
        check(lib.revs_op_nodefast_finish(M, T, self.ksplit1, ptr(self.f_zt), ptr(self.p0),
                                          ptr(self.gmin), ptr(self.inv_sqrt_n), ptr(self.dnode),
                                          ptr(self.slack), ptr(self.f_stats), st),
              "revs_op_nodefast_finish")
        # nodes without residences have gmin = +inf; a clamp is active iff some slack < 0
        viol, pmax = self.f_stats.cpu().tolist()
        if viol > 1e-9 * max(1.0, pmax):
            return "post"
        check(lib.revs_op_node_apply(M, T, ptr(self.node_ptr), ptr(self.inv_sqrt_n), ptr(self.P_est),
                                     ptr(self.P_sch), ptr(self.G), self.kappa, self.preclamp,
                                     ptr(self.dnode), ptr(self.P_est_new), st), "revs_op_node_apply")
        self.op_iters_hist.append(it)
        self.op_path_hist.append("node")
        self.op_converged = converged
        return True

    # ------------------------------------------------- operator, dual Newton path
    def _dual_phase(self, phase: int, y, use_y: bool, k: int):
        lib, M, T = self.lib, self.M, self.T
        check(lib.revs_op_dual_evaluate(
            phase, M, T, ptr(self.node_ptr), ptr(self.P_est), ptr(self.P_sch), ptr(self.G),
            ptr(self.R64), ptr(self.R64T), ptr(y), int(use_y), self.kappa, self.vlo, self.vhi,
            self.op.newton_kadd, self.ksplit1, ptr(self.d_sl), ptr(self.v_sl), ptr(self.pnq),
            ptr(self.P_est_new), ptr(self.vfull), ptr(self.violw), ptr(self.d_part),
            ptr(self.c_idx[k]), ptr(self.c_cnt[k]), ptr(self.c_val[k]), self.stats_dev[k],
            0.0, ptr(self.tile_cnt), self.stream), "revs_op_dual_evaluate")

    def _dual_home_pass_rows(self, y, sup: int):
        """Phase 1 of an evaluation with d = R^T y / kappa taken from the few rows listed in
        candidate set `sup` (they include every row with y != 0) instead of a dense product."""
        check(self.lib.revs_op_dual_eval_rows(
            self.M, self.T, ptr(self.node_ptr), ptr(self.P_est), ptr(self.P_sch), ptr(self.G),
            ptr(self.R64), ptr(self.c_idx[sup]), ptr(self.c_cnt[sup]), ptr(y), self.kappa,
            ptr(self.pnq), ptr(self.P_est_new), self.stream), "revs_op_dual_eval_rows")

    def _dual_launch(self, y, use_y: bool, k: int, full: bool = True, sup=None, record=True):
        """Enqueue one evaluation: p, N, D and the voltage rows for the multipliers y;
        candidate lists and stats into buffer set k, stats on their way to pinned host
        memory.  Also writes P_est_new = max(g0 - R^T y / kappa, 0).  Does not wait.
        One host call; two around the all-reduce of pnq when residences are sharded.  With
        full=False only p is exchanged (enough to judge the voltage rows: the steady-state
        case); N and the dual value follow through _dual_complete if the solve goes on.
        (Replaying the launches as a hipGraph was measured and is slower than issuing
        them: 44 vs 37 us per evaluation.)"""
        if use_y and sup is not None:            # few multipliers: no dense product for d
            self._dual_home_pass_rows(y, sup)
            if self.group is not None:
                self._allreduce(self.pnq if full else self.pnq[0])
            self._dual_phase(2, y, use_y, k)
        elif self.group is None:
            self._dual_phase(3, y, use_y, k)
        else:
            self._dual_phase(1, y, use_y, k)
            self._allreduce(self.pnq if full else self.pnq[0])   # the only exchange
            self._dual_phase(2, y, use_y, k)
        if record and self.stats_ev[k] is not None:
            self.stats_ev[k].record()

    def _dual_complete(self, y, use_y: bool, k: int):
        """After a full=False evaluation that did not settle the solve: exchange N and the
        dual value too and redo the row bookkeeping (the stats' D_t needs the global sum)."""
        if self.group is None:
            return self._dual_wait(k)
        self._allreduce(self.pnq[1:])
        self._dual_phase(2, y, use_y, k)
        if self.stats_ev[k] is not None:
            self.stats_ev[k].record()
        return self._dual_wait(k)

    def _dual_wait(self, k: int):
        if self.stats_ev[k] is not None:
            self.stats_ev[k].synchronize()
        return self.stats_host[k].numpy().copy()

    def _dual_evaluate(self, y, use_y: bool, k: int, sup=None):
        self._dual_launch(y, use_y, k, sup=sup)
        return self._dual_wait(k)

    def _operator_solve_newton(self, first=None, pre=None):
        """Utility.solve through the dual (see csrc/newton_kernels.hip).  True when
        P_est_new holds the answer to tolerance; False hands the iteration to ADMM.
        `first`: stats of an evaluation of the current multipliers already made (buffer 0).
        `pre`: stats of the evaluation (buffer 1) that `_chain_launch` enqueued behind `first`
        without reading it -- small model on candidate set 0, a full step for the slots not
        yet within tolerance -- i.e. the first line-search trial of the first Newton iteration
        if that iteration turns out to be the one this loop would have run.  Sets
        `_pre_kept`: the accepted state is exactly the one `pre` (or `first`) left behind."""
        o, lib, M, T, st = self.op, self.lib, self.M, self.T, self.stream
        A = _lib.DUAL_AMAX
        scale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        ycur, ytrial = self.yd
        cur = 0
        # (`_sup`: a candidate set that lists every row of the current multipliers, while there
        # are few enough of them for the row-wise home pass)
        stt = (self._dual_evaluate(ycur, self._y_support, cur, sup=self._sup)
               if first is None else first)
        evals, newton, pivots, ok_all = 1, 0, 0, False
        best, stall = np.inf, 0
        last_small = False
        from_pre = pre is not None       # P_est_new is what `pre` (== `first` if nothing moved) wrote
        while True:
            if (stt[:, 2] > A).any():
                break                                    # more multipliers than a model holds
            rmax = stt[:, 0] / scale
            if rmax.max() <= o.eps:
                ok_all = True
                break
            if newton >= o.newton_max:
                break
            # a slot whose model is full of multipliers while rows are still violated cannot
            # take them in; and a solve that stopped improving is not worth more iterations
            if ((stt[:, 2] >= A) & (stt[:, 3] > 0) & (rmax > o.eps)).any():
                break
            if rmax.max() < 0.5 * best:
                best, stall = rmax.max(), 0
            else:
                stall += 1
                if stall >= 10:
                    break
            newton += 1
            # model of every slot: K_t = R_F N_t R_F^T / kappa over its candidates, maximised
            # over the sign constraints (block principal pivoting, one workgroup per slot)
            # (candidates of a slot = its rows with a multiplier + the violated rows admitted)
            ncand = stt[:, 2] + np.minimum(stt[:, 3], np.minimum(o.newton_kadd, A - stt[:, 2]))
            self.model_calls[0 if ncand.max() <= 8 else 1] += 1
            last_small = bool(ncand.max() <= 8)
            few = stt[:, 2].max() + o.newton_kadd <= 48
            # (the chain guessed how its trial's home pass gets d = R^T y / kappa -- row-wise or
            # dense; another choice here would differ in the last bits: then redo the trial)
            use_pre = (pre is not None and newton == 1 and last_small
                       and bool(few) == self._chain_few)
            if use_pre:
                pass                             # (the chain ran this model on this set)
            elif ncand.max() <= 8:               # the binding steady state: one small kernel
                check(lib.revs_op_dual_model_small(M, T, ptr(self.R64), ptr(self.pnq[1]),
                                                   ptr(self.c_idx[cur]), ptr(self.c_cnt[cur]),
                                                   ptr(self.c_val[cur]), self.kappa, o.newton_delta,
                                                   o.newton_pivots, ptr(self.k_full), ptr(self.yhat),
                                                   self.info_dev, st), "revs_op_dual_model_small")
            else:
                check(lib.revs_op_dual_model(M, T, ptr(self.R64), ptr(self.pnq[1]),
                                             ptr(self.c_idx[cur]), ptr(self.c_cnt[cur]),
                                             ptr(self.c_val[cur]), self.kappa, o.newton_delta,
                                             o.newton_pivots, self.nks, ptr(self.k_slabs),
                                             ptr(self.k_full), ptr(self.yhat), self.info_dev, st),
                      "revs_op_dual_model")
            D = stt[:, 1]
            pending = rmax > o.eps
            alpha = pending.astype(np.float64)
            nxt = 1 - cur
            for ls in range(o.newton_ls):
                if use_pre and ls == 0:
                    stn = pre                    # that trial and its evaluation: already there
                else:
                    from_pre = False
                    self.alpha_h.numpy()[:] = alpha  # read by the step kernel through its mapping
                    ytrial.copy_(ycur)
                    check(lib.revs_op_dual_step(T, ptr(self.c_idx[cur]), ptr(self.c_cnt[cur]),
                                                ptr(self.c_val[cur]), ptr(self.yhat),
                                                self.alpha_dev, ptr(ytrial),
                                                self.stats_dev[nxt] + 32, st),
                          "revs_op_dual_step")
                    stn = self._dual_evaluate(ytrial, True, nxt, sup=cur if few else None)
                evals += 1
                okk = stn[:, 1] >= D + 1e-4 * stn[:, 4] - 1e-13 * np.abs(D)
                pending &= ~okk
                if not pending.any():
                    break
                alpha[pending] *= 0.5
            pivots += int(np.abs(self.info_h.numpy()).sum())     # (the evaluation was waited for)
            if pending.any():
                break                                    # no ascent found: leave it to ADMM
            ycur, ytrial = ytrial, ycur
            cur, stt = nxt, stn
        self.yd = [ycur, ytrial]
        self.newton_hist.append((newton, evals, pivots))
        self._pre_kept = bool(ok_all and from_pre and newton <= 1)
        # a solve of exactly one Newton iteration on the small model tends to repeat: the next
        # one is enqueued whole (_chain_launch)
        self._chain_ok = bool(ok_all and newton == 1 and last_small and evals == 2)
        self._chain_few = bool(few) if newton >= 1 else False
        # speculate on the next iteration after a solve that needed no Newton iteration -- but
        # after a discarded sweep only once 2, 4, ... 64 such solves have gone by (rows that
        # keep moving in and out of their limits would otherwise cost a wasted sweep each time)
        if ok_all and newton == 0:
            self._spec_wait = max(self._spec_wait - 1, 0)
            self._spec_ok = self._spec_wait == 0
        else:
            self._spec_ok = False
        if not ok_all:
            ycur.zero_()
            self._y_support = False
            self._sup = None
            return False
        self._y_support = bool(stt[:, 2].sum() > 0)
        # the accepted evaluation's candidate set `cur` lists the rows with y != 0 first
        self._sup = cur if (self._y_support and stt[:, 2].max() + o.newton_kadd <= 48) else None
        self.op_iters_hist.append(evals)
        self.op_path_hist.append("dual")
        self.op_converged = True
        return True

    def _chain_launch(self, write_sc, rec):
        """The binding steady state without the host in the loop: evaluation of the current
        multipliers (set 0) with its selection, the small model and the step in one launch
        (full step for the slots that evaluation leaves pending, decided on the device:
        revs_op_dual_select_model_step), the evaluation of the trial (set 1) and the home
        sweep on its answer -- the trial's candidate selection rides in the sweep's launch --,
        all enqueued; nothing is read."""
        o, lib, M, T, st = self.op, self.lib, self.M, self.T, self.stream
        scale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        ycur, ytrial = self.yd
        # (no event records in the chain: each costs the stream ~6 us; the host polls the
        # sequence tag the last selection writes)
        nb = (M + 31) // 32
        nb = nb if (T <= 32 and nb <= 256) else 0
        use_y = self._y_support
        if use_y and self._sup is not None:       # as _dual_launch, the selection left out
            self._dual_home_pass_rows(ycur, self._sup)
        else:
            self._dual_phase(1, ycur, use_y, 0)
        if self.group is not None:
            self._allreduce(self.pnq)
        self._dual_phase(2 | 4, ycur, use_y, 0)
        # selection, small model and step of every slot in one launch
        check(lib.revs_op_dual_select_model_step(
            M, T, ptr(self.d_part), nb, ptr(ycur), self.vlo, self.vhi, o.newton_kadd,
            ptr(self.vfull), ptr(self.violw), ptr(self.c_idx[0]), ptr(self.c_cnt[0]),
            ptr(self.c_val[0]), self.stats_dev[0], 0.0, ptr(self.R64), ptr(self.pnq[1]), self.kappa,
            o.newton_delta, o.newton_pivots, ptr(self.k_full), ptr(self.yhat), self.info_dev, scale,
            o.eps, ptr(ytrial), self.stats_dev[1] + 32, st), "revs_op_dual_select_model_step")
        if self._chain_few:                       # d = R^T y / kappa from the rows of set 0
            self._dual_home_pass_rows(ytrial, 0)
        else:
            self._dual_phase(1, ytrial, True, 1)
        if self.group is not None:
            self._allreduce(self.pnq)
        self._dual_phase(2 | 4, ytrial, True, 1)  # product and rows; selection: in the sweep
        rec(1)
        self._chain_seq -= 1.0
        check(lib.revs_agent_step_select(
            self.n, T, ptr(self.cost), ptr(self.homes), ptr(self.load), ptr(self.P_est),
            ptr(self.P_est_new), ptr(self.P_sch), ptr(self.G), ptr(self.P_sch_alt), ptr(self.G_alt),
            ptr(self.S) if write_sc else None, ptr(self.Csoc) if write_sc else None,
            ptr(self.diff), ptr(self.dsq), ptr(self.status), ptr(self.pdhg_dual), self.kappa,
            self.mode, C.byref(self.pdhg), M, ptr(self.d_part), ptr(ytrial), self.vlo, self.vhi,
            o.newton_kadd, ptr(self.vfull), ptr(self.violw), ptr(self.c_idx[1]), ptr(self.c_cnt[1]),
            ptr(self.c_val[1]), self.stats_dev[1], self._chain_seq, None, None, None, nb, st),
            "revs_agent_step_select")
        rec(2)

    def _chain_accept(self):
        """Wait for the chain's two evaluations and, if they are the usual outcome -- one
        Newton iteration on the small model, full step accepted, converged -- do the
        bookkeeping _operator_solve_newton would do for it (revs_newton_chain_accept makes
        the same checks in one native call).  False: nothing was changed."""
        o = self.op
        if self.stats_ev[0] is not None:
            # the trial's verdict is written early in the sweep's launch: poll its sequence tag
            # (pinned memory) rather than wait for the sweep; stream order puts everything the
            # chain wrote before it
            tags = self.stats_host[1].numpy()[:, 5]
            spins, t0 = 0, None
            while not (tags == self._chain_seq).all():
                spins += 1
                if spins & 0xFFF == 0:
                    import time
                    t0 = t0 or time.monotonic()
                    if time.monotonic() - t0 > 120.0:
                        raise _lib.RevsError("chained Newton iteration: timed out waiting for "
                                             "the evaluation's sequence tag")
        nsum, nmax = C.c_int32(), C.c_int32()
        scale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        if not self.lib.revs_newton_chain_accept(
                self.T, self.stats_host[0].data_ptr(), self.stats_host[1].data_ptr(), scale, o.eps,
                _lib.DUAL_AMAX, o.newton_kadd, int(self._chain_few), C.addressof(nsum),
                C.addressof(nmax)):
            return False
        self._chain_book(nsum.value, nmax.value)
        return True

    def _chain_book(self, nsum, nmax):
        o = self.op
        self.yd = [self.yd[1], self.yd[0]]
        self.model_calls[0] += 1
        self.newton_hist.append((1, 2, int(np.abs(self.info_h.numpy()).sum())))
        self._pre_kept, self._chain_ok, self._spec_ok = True, True, False
        self._y_support = nsum > 0
        self._sup = 1 if (self._y_support and nmax + o.newton_kadd <= 48) else None
        self.op_iters_hist.append(2)
        self.op_path_hist.append("dual")
        self.op_converged = True

    def _require_converged(self, ok):
        """An operator answer that did not reach its tolerance is not handed to the residences:
        REVS_ENOTCONV (every rank holds the same node-space state and takes the same decision)."""
        if not ok:
            raise _lib.RevsError(
                "REVS_ENOTCONV: the operator QP (lpsolver.py:163-238) did not converge -- the dual "
                f"Newton path gave up (more than {_lib.DUAL_AMAX} binding rows in a slot, no ascent, or "
                f"{self.op.newton_max} iterations) and the ADMM form stopped at max_iter = "
                f"{self.op.max_iter} above eps = {self.op.eps:g}")

    def operator_solve(self, admm_only=False):
        self._p_clear = None
        """Utility(graph, P_est[k], P_sch[k], G[k]).solve() -> P_est[k+1]
        (lpsolver.py:256-259), written to self.P_est_new.  `admm_only`: skip the dual
        Newton attempt (the caller has just seen it fail for this state)."""
        o, lib, M, T, st = self.op, self.lib, self.M, self.T, self.stream
        self._fused_ready = False
        if o.solver == "newton" and not admm_only:
            if self._operator_solve_newton():
                return True
            self._fast_cold = True
            self.op_cold = True
        self._ensure_admm()
        if self._fast_ok:
            if self._fast_wait > 0:
                self._fast_wait -= 1
            else:
                r = self._operator_solve_node(precheck=self._fast_backoff > 1)
                if r is True:
                    self._fast_backoff = 1
                    return self.op_converged
                # Some residence would go to zero: the g >= 0 rows matter for this state, so
                # the general home-space ADMM solves it.  The fast path is tried again later
                # (clamps are typical of the first ADMM iterations and often disappear): at
                # once when only the free pre-check g0 >= 0 failed, after an exponentially
                # growing number of iterations when a whole fast solve was wasted.
                if r == "post":
                    self._fast_backoff = min(2 * self._fast_backoff, 64)
                    self._fast_wait = self._fast_backoff
                self._fast_cold = True
                self.op_cold = True
        _lib.check(lib.revs_op_g0(self.n, T, ptr(self.P_est), ptr(self.P_sch), ptr(self.G),
                                  self.kappa, ptr(self.g0), st), "revs_op_g0")
        if self.op_cold or not o.warm_start:
            _lib.check(lib.revs_op_init_home(self.n, T, ptr(self.g0), ptr(self.sb), st),
                       "revs_op_init_home")
            _lib.check(lib.revs_aggregate_f64(M, T, ptr(self.node_ptr), ptr(self.sb),
                                              ptr(self.inv_sqrt_n), ptr(self.rhat), st),
                       "revs_aggregate_f64")
            self._allreduce(self.rhat)
            self._gemm(self.Q, self.rhat, self.ta[0])
            _lib.check(lib.revs_op_row_scale(M, T, ptr(self.s), ptr(self.ta[0]),
                                             ptr(self.tb[0]), st), "revs_op_row_scale")
            self._gemm(self.QT, self.tb[0], self.cx)                # cx = C_v x, x = max(g0,0)
            self.rho_v.fill_(o.rho_v_scale * self.kappa / self.smax ** 2)
            self.rho_b.fill_(o.rho_b_scale * self.kappa)
            _lib.check(lib.revs_op_init_node(M, T, ptr(self.cx), ptr(self.rho_v),
                                             ptr(self.sqrt_n), self.vlo, self.vhi, ptr(self.zv),
                                             ptr(self.yv), ptr(self.w), st), "revs_op_init_node")
            self.op_cold = False
        self._home_pass(with_update=False)
        it, converged = 0, False
        while it < o.max_iter:
            self._inner_block()
            it += o.check_every
            rel_p, rel_d = self._rel_residuals(self._residuals())
            if max(rel_p.max(), rel_d.max()) <= o.eps:
                converged = True
                break
            if o.calibrate and not self._calibrated:
                it += self._calibrate_rho()
                continue
            if o.adapt_every and it % o.adapt_every == 0:
                sc = np.sqrt(np.maximum(rel_p, 1e-14) / np.maximum(rel_d, 1e-14))
                sc = np.clip(sc, 0.2, 5.0)
                sc = np.where((sc > 2.0) | (sc < 0.5), sc, 1.0)
                if (sc != 1.0).any():
                    sct = torch.from_numpy(sc).to(self.dev)
                    self.rho_v.mul_(sct)
                    self.rho_b.mul_(sct)
                    # y_b = min(s_b,0) is kept as is; only w and rhat depend on rho
                    _lib.check(lib.revs_op_node_w(M, T, ptr(self.zv), ptr(self.yv),
                                                  ptr(self.rho_v), ptr(self.w), st),
                               "revs_op_node_w")
                    self._home_pass(with_update=False)
        self.op_iters_hist.append(it)
        self.op_path_hist.append("home")
        self.op_converged = converged
        _lib.check(lib.revs_op_export(self.n, T, ptr(self.sb), ptr(self.P_est_new), st),
                   "revs_op_export")
        return converged

    # ----------------------------------------------------------------- homes
    def agent_step(self, write_sc=True, to_alt=False):
        """All Home(...).solve() of one iteration + dual update + diff
        (lpsolver.py:269-284) in one kernel launch.  With `to_alt` P_sch[k+1] and G[k+1] go
        to the spare buffers (speculative launch, see step)."""
        self._fused_ready = False
        ps_out, g_out = (self.P_sch_alt, self.G_alt) if to_alt else (self.P_sch, self.G)
        check(self.lib.revs_agent_step_out(
            self.n, self.T, ptr(self.cost), ptr(self.homes), ptr(self.load), ptr(self.P_est),
            ptr(self.P_est_new), ptr(self.P_sch), ptr(self.G), ptr(ps_out), ptr(g_out),
            ptr(self.S) if write_sc else None, ptr(self.Csoc) if write_sc else None,
            ptr(self.diff), ptr(self.dsq), ptr(self.status), ptr(self.pdhg_dual),
            self.kappa, self.mode,
            C.byref(self.pdhg), self.stream), "revs_agent_step_out")

    def residuals(self, eps=1e-4):
        """Residuals of the iteration just finished, reduced on the device:
        (|P_est - P_sch|_2, kappa |dP_sch|_2, max_h diff[h], converged) where converged
        means max_h diff[h] <= eps -- diff (lpsolver.py:284) is the reference's only
        convergence measure."""
        check(self.lib.revs_residual_finalize(ptr(self.diff), ptr(self.dsq), self.n, self.T,
                                              self.kappa, eps, ptr(self.res_scratch),
                                              ptr(self.resid), self.stream),
              "revs_residual_finalize")
        r = self.resid.cpu().numpy().astype(np.float64)
        self.check_status()
        if self.group is not None:
            t = torch.tensor([r[0] ** 2, r[1] ** 2], dtype=torch.float64, device=self.dev)
            self._allreduce(t)
            mx = torch.tensor([r[2]], dtype=torch.float64, device=self.dev)
            self._allreduce(mx, torch.distributed.ReduceOp.MAX)
            return (math.sqrt(t[0].item()), math.sqrt(t[1].item()), mx.item(),
                    bool(mx.item() <= eps))
        return r[0], r[1], r[2], bool(r[3] > 0.5)

    def step(self, write_sc=True, events=None):
        """One iteration of the while-loop of lpsolver.py:254-287.  `events`: HIP events
        [1], [2] recorded between the operator part and the home sweep, and after the sweep
        (bench.py's per-kernel timing; [0] is the caller's: the previous step's [2])."""
        rec = (lambda i: events[i].record()) if events else (lambda i: None)
        o = self.op
        if not write_sc and not events and self._fused_ready and self._stream_ok():
            self._stream_run(1)              # steady state: one launch, verdict inside it
            return
        self._p_clear = None                 # (every other path rewrites the node-sum arrays)
        if o.solver == "newton" and o.speculate and self._spec_ok:
            # steady state: the multipliers of the last iteration are expected to stand
            scale = max(abs(self.vlo), abs(self.vhi), 1e-300)
            if self._plan is not None:               # one native call: enqueue, wait, judge
                rm = C.c_double()
                evh = [None, None]
                if events:
                    for i in (1, 2):
                        if not events[i].cuda_event:
                            events[i].record()       # creates the hipEvent behind the object
                        evh[i - 1] = events[i].cuda_event
                fused_in = self._fused_ready
                p0 = self.pnq[0]
                p_in = self._fused_p if fused_in else p0
                p_out = None
                if o.fuse_home_pass and not self._y_support:
                    p_out = self.p_alt if p_in.data_ptr() == p0.data_ptr() else p0
                self._fused_ready = False

                def call(phase):
                    check(self.lib.revs_plan_spec_step(
                        self._plan, phase, ptr(self.yd[0]), int(self._y_support), ptr(self.P_est),
                        ptr(self.P_est_new), ptr(self.P_sch), ptr(self.G), ptr(self.P_sch_alt),
                        ptr(self.G_alt), ptr(self.S) if write_sc else None,
                        ptr(self.Csoc) if write_sc else None, int(fused_in), ptr(p_in), ptr(p_out),
                        ptr(self.P_est_alt), C.byref(rm), evh[0], evh[1],
                        self.stream), "revs_plan_spec_step")
                ar_ahead = prod_ahead = False
                skip_product = fused_in and self._ar_ahead and self._prod_ahead
                if self.group is None:
                    call(3)
                else:                                # home pass, exchange of p, the rest
                    if not fused_in:
                        call(1)
                    if not (fused_in and self._ar_ahead):
                        self._allreduce(p_in)        # the only exchange of the iteration
                    self._ar_ahead = self._prod_ahead = False
                    if p_out is None:
                        call(2 | (4 if skip_product else 0))
                    else:
                        # enqueue product and sweep, then -- before waiting for the verdict --
                        # the exchange of the node sums this sweep leaves for the NEXT
                        # evaluation: it is stream-ordered behind the sweep, and its host-side
                        # cost overlaps the sweep instead of standing between two iterations
                        # (a discarded sweep makes it a wasted, harmless exchange; every rank
                        # takes the same decisions, so the collectives stay in step)
                        call(2 | 16 | (4 if skip_product else 0))
                        self._allreduce(p_out)
                        ar_ahead = True
                        if fused_in:
                            # ... and the next product behind it (not after an evaluation whose
                            # stats a discard would continue from: its node sums must survive)
                            check(self.lib.revs_plan_spec_step(
                                self._plan, 64, ptr(self.yd[0]), 0, None, None, None, None, None, None,
                                None, None, 0, ptr(p_out), ptr(p_in), None, None, None, None,
                                self.stream), "revs_plan_spec_step")
                            prod_ahead = True
                        call(32)
                fuse_out = p_out is not None
                kept = rm.value / scale <= o.eps
                # after a fused home pass the stats carry no dual value: a discarded sweep is
                # followed by a fresh evaluation instead of a continuation from these stats
                stt = None if (kept or fused_in) else self.stats_host[0].numpy().copy()
                if kept and fuse_out:
                    self._fused_ready, self._fused_p = True, p_out
                    self._ar_ahead, self._prod_ahead = ar_ahead, prod_ahead
            else:
                self._dual_launch(self.yd[0], self._y_support, 0, full=False)
                rec(1)
                self.agent_step(write_sc, to_alt=True)
                rec(2)
                stt = self._dual_wait(0)
                kept = stt[:, 0].max() / scale <= o.eps
            if kept:
                self.P_sch, self.P_sch_alt = self.P_sch_alt, self.P_sch
                self.G, self.G_alt = self.G_alt, self.G
                self.op_iters_hist.append(1)
                self.op_path_hist.append("dual")
                self.newton_hist.append((0, 1, 0))
                self.op_converged = True
                self.spec_hist[0] += 1
                self._spec_back = 1
            else:                          # rows need work: finish the solve, redo the sweep
                self._spec_discard(stt, write_sc)
        elif o.solver == "newton" and o.chain and self._chain_ok:
            # binding steady state: the last solve was one Newton iteration on the small model;
            # enqueue the same again, and the sweep behind it, before reading anything
            self._fused_ready = False
            if self._plan is not None and self.group is None:
                # one native call: the six launches, the wait and the verdict
                acc, nsum, nmax = C.c_int32(), C.c_int32(), C.c_int32()
                evh = [None, None]
                if events:
                    for i in (1, 2):
                        if not events[i].cuda_event:
                            events[i].record()
                        evh[i - 1] = events[i].cuda_event
                sup0 = self._sup if (self._y_support and self._sup is not None) else -1
                check(self.lib.revs_plan_chain_step(
                    self._plan, ptr(self.yd[0]), ptr(self.yd[1]), int(self._y_support), sup0,
                    int(self._chain_few), ptr(self.P_est), ptr(self.P_est_new), ptr(self.P_sch),
                    ptr(self.G), ptr(self.P_sch_alt), ptr(self.G_alt),
                    ptr(self.S) if write_sc else None, ptr(self.Csoc) if write_sc else None,
                    C.addressof(acc), C.addressof(nsum), C.addressof(nmax), evh[0], evh[1],
                    self.stream), "revs_plan_chain_step")
                self._chain_finish(bool(acc.value), nsum.value, nmax.value, write_sc)
            else:
                self._chain_launch(write_sc, rec)
                # (_chain_accept books the usual outcome itself)
                if self._chain_accept():
                    self.P_sch, self.P_sch_alt = self.P_sch_alt, self.P_sch
                    self.G, self.G_alt = self.G_alt, self.G
                    self.chain_hist[0] += 1
                else:
                    self._chain_finish(False, 0, 0, write_sc)
        else:
            self._fused_ready = False
            self._require_converged(self.operator_solve())
            rec(1)
            self.agent_step(write_sc)
            rec(2)
        self.P_est, self.P_est_new = self.P_est_new, self.P_est
        if self._fused_ready:             # the next P_est_new is already in the spare buffer
            self.P_est_new, self.P_est_alt = self.P_est_alt, self.P_est_new
        self.iteration += 1

    def _spec_discard(self, stt, write_sc):
        """A speculative sweep whose evaluation found rows beyond tolerance: finish the Newton
        solve (from the evaluation's stats `stt` when they carry a dual value), run the sweep
        on its answer; speculation backs off."""
        self.spec_hist[1] += 1
        self._spec_back = min(2 * self._spec_back, 64)
        self._spec_wait = self._spec_back
        if stt is not None:
            stt = self._dual_complete(self.yd[0], self._y_support, 0)
        if not self._operator_solve_newton(first=stt):
            self._fast_cold = True
            self.op_cold = True
            self._require_converged(self.operator_solve(admm_only=True))
        self.agent_step(write_sc)

    def _chain_run(self, count):
        """Up to `count` iterations of the binding steady state inside one native call
        (revs_plan_chain_run); the first one that is not the usual outcome is finished here as
        step() would.  Returns the number of iterations done (at least one)."""
        self._fused_ready = False
        self._p_clear = None
        ys = (self.yd[0], self.yd[1])
        bufs = (self.P_est, self.P_est_new, self.P_sch, self.P_sch_alt, self.G, self.G_alt)
        sup0 = self._sup if (self._y_support and self._sup is not None) else -1
        st = _lib.ChainState(ptr(ys[0]), ptr(ys[1]), int(self._y_support), sup0,
                             *[ptr(t) for t in bufs])
        kept = C.c_int32()
        check(self.lib.revs_plan_chain_run(self._plan, count, C.byref(st), int(self._chain_few),
                                           C.addressof(kept), self.stream), "revs_plan_chain_run")
        n = kept.value
        by = {t.data_ptr(): t for t in bufs}
        self.P_est, self.P_est_new = by[st.p_est], by[st.p_est_new]
        self.P_sch, self.P_sch_alt = by[st.p_sch], by[st.p_sch_alt]
        self.G, self.G_alt = by[st.gamma], by[st.gamma_alt]
        self.yd = [ys[0], ys[1]] if st.y == ys[0].data_ptr() else [ys[1], ys[0]]
        if n:
            self.model_calls[0] += n
            self.newton_hist.extend([(1, 2, -1)] * n)      # (pivot counts not read)
            self._pre_kept, self._chain_ok, self._spec_ok = True, True, False
            self._y_support = bool(st.use_y)
            self._sup = 1 if st.sup0 == 1 else None
            self.op_iters_hist.extend([2] * n)
            self.op_path_hist.extend(["dual"] * n)
            self.op_converged = True
            self.chain_hist[0] += n
            self.iteration += n
        if n == count:
            return n
        # the call stopped at an iteration for the general loop (its launches are made)
        self._chain_finish(False, 0, 0, False)
        self.P_est, self.P_est_new = self.P_est_new, self.P_est
        self.iteration += 1
        return n + 1

    def _chain_finish(self, accepted, nsum, nmax, write_sc):
        """After the chain's launches: book the usual outcome, or hand both evaluations to the
        general loop (which reuses the trial where it is exactly its own first step); keep the
        speculative sweep or run it again."""
        if accepted:
            self._chain_book(nsum, nmax)
            ok = True
        else:                                # (the tag was seen: both blocks are complete)
            stt0, stn = (self.stats_host[0].numpy().copy(), self.stats_host[1].numpy().copy())
            ok = self._operator_solve_newton(first=stt0, pre=stn)
        if ok and self._pre_kept:
            self.P_sch, self.P_sch_alt = self.P_sch_alt, self.P_sch
            self.G, self.G_alt = self.G_alt, self.G
            self.chain_hist[0] += 1
        else:
            self.chain_hist[1] += 1
            if not ok:
                self._fast_cold = True
                self.op_cold = True
                self._require_converged(self.operator_solve(admm_only=True))
            self.agent_step(write_sc)

    def run_steps(self, count):
        """`count` iterations of step(write_sc=False).  Consecutive steady-state iterations
        (one GPU, no multipliers, speculation on) run inside ONE native call,
        revs_plan_spec_run -- the buffer rotation included, no Python between the launches --
        which returns at the first sweep that has to be discarded; that iteration is finished
        here as step() would, and the loop goes on.  Same trajectory as calling step()."""
        o, done = self.op, 0
        scale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        while done < count:
            if self._stream_ok():
                if not self._fused_ready:            # entry: one step of the general driver
                    self.step(write_sc=False)
                    done += 1
                else:
                    done += self._stream_run(count - done)
                continue
            native = self._plan is not None and self.group is None and o.solver == "newton"
            if native and o.chain and self._chain_ok and not (o.speculate and self._spec_ok):
                done += self._chain_run(count - done)
                continue
            if not (native and o.speculate and self._spec_ok and o.fuse_home_pass
                    and not self._y_support):
                self.step(write_sc=False)
                done += 1
                continue
            p0 = self.pnq[0]
            self._p_clear = None
            if self._fused_ready and self._fused_p is self.p_alt2:     # (revs_plan_spec_run knows two arrays)
                p0.copy_(self.p_alt2)
                self._fused_p = p0
            bufs = (self.P_est, self.P_est_new, self.P_est_alt, self.P_sch, self.P_sch_alt, self.G,
                    self.G_alt)
            st = _lib.SpecState(*[ptr(t) for t in bufs], ptr(p0), ptr(self.p_alt),
                                ptr(self._fused_p) if self._fused_ready else None,
                                int(self._fused_ready))
            kept, fin, rm = C.c_int32(), C.c_int32(), C.c_double()
            check(self.lib.revs_plan_spec_run(self._plan, count - done, ptr(self.yd[0]), C.byref(st),
                                              scale, o.eps, C.addressof(kept), C.addressof(fin),
                                              C.addressof(rm), self.stream), "revs_plan_spec_run")
            n = kept.value
            by = {t.data_ptr(): t for t in bufs}
            self.P_est, self.P_est_new, self.P_est_alt = (by[st.p_est], by[st.p_est_new],
                                                          by[st.p_est_alt])
            self.P_sch, self.P_sch_alt = by[st.p_sch], by[st.p_sch_alt]
            self.G, self.G_alt = by[st.gamma], by[st.gamma_alt]
            self._fused_ready = bool(st.fused_ready)
            if self._fused_ready:
                self._fused_p = p0 if st.fused_p == p0.data_ptr() else self.p_alt
            if n:
                self.op_iters_hist.extend([1] * n)
                self.op_path_hist.extend(["dual"] * n)
                self.newton_hist.extend([(0, 1, 0)] * n)
                self.op_converged = True
                self.spec_hist[0] += n
                self._spec_back = 1
                self.iteration += n
                done += n
            if done < count and n < count - (done - n):
                # the call stopped at a sweep to discard: finish that iteration as step() does
                fused_in = bool(fin.value)
                self._fused_ready = False
                stt = None if fused_in else self.stats_host[0].numpy().copy()
                self._spec_discard(stt, False)
                self.P_est, self.P_est_new = self.P_est_new, self.P_est
                self.iteration += 1
                done += 1

    def _stream_ok(self):
        """The steady state as one launch per iteration (revs_plan_stream_run) applies: a plan
        with the feeder's tree, no multipliers, speculation allowed, and -- sharded -- the
        library's own communicator."""
        o = self.op
        return (self._plan is not None and self._tree is not None and o.solver == "newton"
                and o.speculate and self._spec_ok and o.fuse_home_pass and not self._y_support
                and (self.group is None or self._comm is not None))

    def _stream_run(self, count):
        """Up to `count` steady-state iterations (at most the current burst), one launch each,
        enqueued in one go by the native loop; the first iteration whose rows are not within
        tolerance silences the launches behind it and is finished here as step() finishes a
        discarded speculative sweep.  Returns the number of iterations done (at least one)."""
        o = self.op
        scale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        p0 = self._fused_p
        rest = [b for b in (self.pnq[0], self.p_alt, self.p_alt2) if b.data_ptr() != p0.data_ptr()]
        # p[1] must be zero on entry.  The last launch of a fully kept call has just cleared the
        # array that is p[1] now (roles rotate): no fill launch then
        if self._p_clear is not None and self._p_clear.data_ptr() != p0.data_ptr():
            rest = [self._p_clear] + [b for b in rest if b.data_ptr() != self._p_clear.data_ptr()]
        else:
            rest[0].zero_()
        self._p_clear = None
        if self.group is not None and not self._ar_ahead:
            self._allreduce(p0)
        pes = (self.P_est, self.P_est_new, self.P_est_alt)
        pss, gs, ps = (self.P_sch, self.P_sch_alt), (self.G, self.G_alt), (p0, rest[0], rest[1])
        st = _lib.StreamState()
        for i in range(3):
            st.p_est[i], st.p[i] = ptr(pes[i]), ptr(ps[i])
        for i in range(2):
            st.p_sch[i], st.gamma[i] = ptr(pss[i]), ptr(gs[i])
        kept, rm = C.c_int32(), C.c_double()
        count = min(count, self._burst)
        check(self.lib.revs_plan_stream_run(self._plan, count, C.byref(st), scale, o.eps,
                                            C.addressof(kept), C.addressof(rm), self.stream),
              "revs_plan_stream_run")
        n = kept.value
        self._burst = min(4 * self._burst, o.stream_burst_max) if n == count else o.stream_burst
        by = {t.data_ptr(): t for t in pes + pss + gs + ps}
        self.P_est, self.P_est_new, self.P_est_alt = (by[st.p_est[i]] for i in range(3))
        self.P_sch, self.P_sch_alt = by[st.p_sch[0]], by[st.p_sch[1]]
        self.G, self.G_alt = by[st.gamma[0]], by[st.gamma[1]]
        self._fused_p = by[st.p[0]]
        self._p_clear = by[st.p[1]] if n == count else None
        self._prod_ahead = False
        self._ar_ahead = self.group is not None
        if n:
            self.op_iters_hist.extend([1] * n)
            self.op_path_hist.extend(["dual"] * n)
            self.newton_hist.extend([(0, 1, 0)] * n)
            self.op_converged = True
            self.spec_hist[0] += n
            self._spec_back = 1
            self.iteration += n
        if n == count:
            return n
        # iteration n's verdict failed (its sweep wrote to the spares only; every launch behind
        # it was a no-op): finish it as step() does for a discarded speculative sweep
        self._fused_ready = False
        self._spec_discard(None, False)
        self.P_est, self.P_est_new = self.P_est_new, self.P_est
        self.iteration += 1
        return n + 1

    def check_status(self):
        """Raise if a residence reported 'no solution' (lpsolver.py:153-155) or a PDHG residence
        stopped at its iteration cap, in any sweep since the last check -- a collective decision
        when residences are sharded (every rank raises, or none).  Call at a point where the
        stream has been synchronised (run(), result() and residuals() do)."""
        if self._plan is not None:
            f = int(self.lib.revs_plan_status_flags(self._plan, 1))
        else:
            f = 0
        f |= int((self.status & 3).max().item()) if self.n else 0
        if self.group is not None:
            t = torch.tensor([float(f & 1), float((f >> 1) & 1)], dtype=torch.float64, device=self.dev)
            self._allreduce(t, torch.distributed.ReduceOp.MAX)
            a, b = t.cpu().tolist()
            f = int(a) | (int(b) << 1)
        if f & 1:
            raise _lib.RevsError("No solution found (lpsolver.py:153-155): a residence's "
                                 "charging window cannot reach 90% state of charge")
        if f & 2:
            raise _lib.RevsError("REVS_ENOTCONV: a residence's PDHG iteration reached max_iter "
                                 f"({self.pdhg.max_iter}) before its tolerance ({self.pdhg.tol:g})")

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                self.lib.revs_plan_destroy(self._plan)
                self._plan = None
            if getattr(self, "_comm", None):
                self.lib.revs_comm_destroy(self._comm)
                self._comm = None
        except Exception:
            pass

    def run(self, iter_max=15):
        """Full solve_ADMM loop; returns diff (iter_max, n) in the caller's home order."""
        diffs = np.zeros((iter_max, self.n), np.float32)
        for k in range(iter_max):
            self.step(write_sc=(k == iter_max - 1))
            diffs[k] = self.diff.cpu().numpy()[self.inv_perm]      # (synchronises)
            self.check_status()
        return diffs

    # ------------------------------------------------------- state in / out
    def set_state(self, P_est, P_sch, G, iteration=None):
        """Load (P_est[k], P_sch[k], G[k]) -- the whole state of lpsolver.py:254-287 -- in the
        caller's home order: resume a run, or continue from somebody else's iterate.  Whatever
        the engine had prepared ahead for its own state (the next evaluation's home pass and
        its node sums, an exchanged or multiplied copy of them) is dropped; the operator's
        multipliers stay as a warm start."""
        for t, a in ((self.P_est, P_est), (self.P_sch, P_sch), (self.G, G)):
            a = np.ascontiguousarray(np.asarray(a, np.float32)[self.perm])
            assert a.shape == (self.n, self.T)
            t.copy_(torch.from_numpy(a))
        self._fused_ready = self._ar_ahead = self._prod_ahead = False
        self._fused_p = None
        self._p_clear = None
        self._chain_ok = False
        if iteration is not None:
            self.iteration = int(iteration)

    def get_state(self):
        """(P_est[k], P_sch[k], G[k]) in the caller's home order."""
        return self._unsort(self.P_est), self._unsort(self.P_sch), self._unsort(self.G)

    # ----------------------------------------------------------- inspection
    def _unsort(self, t):
        return t.cpu().numpy()[self.inv_perm]

    def result(self):
        """(P_sch, S, C) of the last iteration in the caller's home order."""
        out = self._unsort(self.P_sch), self._unsort(self.S), self._unsort(self.Csoc)
        self.check_status()
        return out

    def voltage(self, profile=None):
        """R . (node aggregate of a home profile) on the f32 matrix cores: the
        operator's voltage-sensitivity check (lpsolver.py:191-193, drawing.py:60-78)."""
        prof = self.P_sch if profile is None else profile
        check(self.lib.revs_aggregate_f32(self.M, self.T, ptr(self.node_ptr), ptr(prof),
                                          ptr(self.node_load), self.stream), "revs_aggregate_f32")
        self._allreduce(self.node_load)
        check(self.lib.revs_voltage_f32(self.M, self.T, ptr(self.R32), ptr(self.node_load),
                                        ptr(self.volt), self.stream), "revs_voltage_f32")
        return self.volt


def residence_solve(tariff, homes, load, device="cuda:0"):
    """solve_residence for every home (lpsolver.py:430-460) -> p, soc, g arrays."""
    lib = _lib.load()
    dev = _dev_check(device)
    load = np.ascontiguousarray(load, np.float32)
    n, T = load.shape
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_t, d_l = up(np.asarray(tariff, np.float32)), up(load)
    d_h = up(homes.view(np.uint8).reshape(n, HOME_DTYPE.itemsize))
    p = torch.zeros(n, T, dtype=torch.float32, device=dev)
    g = torch.zeros_like(p)
    soc = torch.zeros(n, T + 1, dtype=torch.float32, device=dev)
    check(lib.revs_residence_solve(n, T, ptr(d_t), ptr(d_h), ptr(d_l), ptr(p), ptr(soc), ptr(g),
                                   torch.cuda.current_stream(dev).cuda_stream),
          "revs_residence_solve")
    return p.cpu().numpy(), soc.cpu().numpy(), g.cpu().numpy()
