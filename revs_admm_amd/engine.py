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

import contextlib
import ctypes as C
import dataclasses
import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import HOME_DTYPE, MODES, PDHG, check, ptr
from .feeder import feeder_tree, tree_from_R, tree_voltage_host  # noqa: F401  (re-exported)
from .operator_admm import AdmmFormsMixin
from .operator_newton import DualNewtonMixin
from .steady_state import SteadyStateMixin

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
    # violated rows admitted to a slot's model per Newton iteration.  Round 4: 2 (was 6, then 3).  With on/off chargers the
    # binding steady state's slowest slot sees a handful of newly violated rows every ADMM iteration; admitting
    # six made its model 8 x 8 -- 36 Gram sums, an 8 x 8 pivoting problem, six arg-max rounds, six more rows of R --
    # for the same number of Newton steps as admitting two or three (measured: 0.069 -> 0.060 ms per iteration at
    # 100 000 x 24; cold solves on the 121144 feeder: 103 evaluations in 15 iterations against 92, same wall time).
    # 3 -> 2 once the pivoting kernel was faster: binary 0.054 -> 0.052, binding 0.0434 -> 0.0430 ms, the feeder's 15
    # iterations 11.4 -> 10.9 ms, the transient 3.5 -> 3.3 ms.  (1 starves the cold solves: the 121144 feeder's hand
    # themselves to the ADMM forms, 300 ms; 4 and more lengthen the chained iteration's slowest slot.)
    newton_kadd: int = 2
    # ... but newton_kadd_cold of them in an evaluation that follows one in which some slot showed more than
    # newton_kadd_cold_at violated rows without a multiplier and at least half of the rows admitted the time before kept
    # one (round 5; on the synthetic feeders' long laterals three multipliers clear a thousand violated rows and the
    # small lists stay -- tools/newton_trace.py): a cold solve -- the first ADMM iterations of the
    # 121144 feeder end with 50-69 binding rows in a slot -- took as many Newton iterations as half of those rows
    # (19 and 17 in iterations 2 and 3, of 78 in all 15); the warm solves keep their small models.  0: off.
    # Measured on that feeder (tests/tools/feeder_iters.py --kadd-cold, 15 iterations): off 11.1 ms / 98 evaluations;
    # 4 above 2: 10.0; 8 above 4: 10.1; 12 above 6: 9.9; 16 above 6: 9.7 / 78 evaluations; 32 above 8: 10.5.
    newton_kadd_cold: int = 16
    newton_kadd_cold_at: int = 6
    newton_trace: bool = False   # Python Newton loop only (native_newton=False): per evaluation (ADMM iteration, Newton iteration,
                                 # violated rows without a multiplier: sum / max over slots, rows with one: sum / max, kadd) in .newton_trace
    chain: bool = True           # binding steady state: one Newton iteration enqueued unread
    # ... and folded (one GPU, feeder as a tree): the sweep forms the operator's answer for the trial
    # itself and folds both evaluations' node sums into its own pass -- one pass over the residences
    # and two launches per iteration instead of three passes and five (revs_plan_chain_fold_run)
    chain_fold: bool = True
    fold_redo: int = 2           # Newton steps beyond the first taken inside the folded chain per iteration (0: handed back)
    newton_delta: float = 1e-10  # relative diagonal shift of the model Hessian
    newton_pivots: int = 300     # block-pivoting limit per model problem
    newton_ls: int = 30          # Armijo halvings
    # After an operator solve that needed no Newton iteration, the next ADMM iteration
    # launches the home sweep right behind the operator's first evaluation, before the host
    # has seen that evaluation's verdict (P_sch / G go to spare buffers): the GPU never
    # waits for the host.  If rows turn out to need work, the sweep is simply run again.
    # the Newton solve itself (evaluations, models, line search, stopping tests) as one native call where the plan
    # exists (revs_plan_newton_solve: same iterates, no interpreter between the launches); False: the Python loop of
    # operator_newton.py, which is also what runs on a process group without the library's communicator
    native_newton: bool = True
    # the steady-state / chained iterations as native calls on a plan (revs_plan_*); False: every iteration issued from
    # Python (tests compare the two).  recompute_pe_new: with no multipliers the sweep recomputes the operator's answer
    # from the three profiles it is a function of instead of reading it (same bits, one input stream less).
    # library_comm: residences sharded over an nccl group -> the library's own RCCL communicator inside the native
    # loops; False: torch.distributed issues the all-reduces from Python (round 1's form)
    native_plan: bool = True
    recompute_pe_new: bool = True
    library_comm: bool = True
    speculate: bool = True
    # One GPU, multipliers all zero: the speculative sweep also does the home pass of the NEXT
    # operator evaluation (one pass over the homes per ADMM iteration instead of two).
    fuse_home_pass: bool = True
    # How the steady state judges the voltage rows of an estimate: "dense" = the f64 matrix-core
    # product R p (always possible); "tree" = two tree passes over the radial feeder, O(nodes)
    # instead of O(nodes^2), inside the sweep's own launch (needs the feeder: `feeder=` of
    # AdmmEngine); "auto" = tree when a feeder of at most REVS_TREE_MAX (16 384) nodes was given.
    voltage: str = "auto"
    # streaming steady state: launches enqueued per native call.  Starts at stream_burst, x4 after
    # every call whose launches were all kept (up to stream_burst_max), back to stream_burst after
    # a failed verdict: the launches behind a failure are silenced on the device but still cost
    # ~3 us each, so a regime that fails often keeps its bursts short
    stream_burst: int = 8
    stream_burst_max: int = 512
    # The buffers the run loops need beyond the constructor's (state pools of the streaming steady state, the folded
    # chain's arrays, ring and events: revs_plan_prepare) are allocated at construction; False: on first use.
    preallocate: bool = True
    # column slabs the model Hessian's Gram launch splits the nodes into (0: M // 128, at most 16)
    newton_nks: int = 0
    # Residences sharded: the verdicts of this many consecutive iterations are taken together,
    # after ONE all-reduce of their node sums (revs_plan_set_stream_block) -- a collective per
    # sweep would make the collective's latency the step.  1: every iteration, as on one GPU.
    # stream_block_single: blocks on one GPU too (default: the same loop on one GPU and on eight;
    # False = every launch judges itself there, the round-2 form the tests compare against).
    stream_block: int = 32
    stream_block_single: bool = True
    # ... and, with blocks, this many consecutive ADMM iterations per sweep launch: no verdict is
    # needed between them, so every residence's state stays in registers (revs_agent_step_multi;
    # profiles read and written once per stream_inner iterations, same bits as one launch each)
    stream_inner: int = 32       # (the library bounds it by the launch's LDS: 32 up to T = 24, 16 up to 96, 8 up to 192)
    # ... with the all-reduce and the verdicts of a block on a second stream, beside the sweeps of
    # the next block (the collective is hidden as long as it is shorter than a block of sweeps)
    stream_overlap: bool = True


def _dev_check(device):
    if not torch.cuda.is_available():
        raise _lib.RevsError("revs_admm_amd needs a ROCm GPU (torch.cuda.is_available() is "
                             "False); there is no CPU fallback")
    return torch.device(device)


def _on_current_stream(fn):
    """step / run_steps / run: torch's current stream, looked up once at the outermost entry."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        if self._stream_pin is not None or self.dev.type != "cuda":
            return fn(self, *a, **k)
        self._stream_pin = torch.cuda.current_stream(self.dev).cuda_stream
        try:
            return fn(self, *a, **k)
        finally:
            self._stream_pin = None
    return wrapped


class AdmmEngine(DualNewtonMixin, AdmmFormsMixin, SteadyStateMixin):
    """State of one ADMM run on one GPU.

    Parameters
    ----------
    cost      (T,)   tariff
    homes     (n,)   HOME_DTYPE records (pack_homes) -- this rank's residences
    load      (n,T)  base load
    node_of   (n,)   constraint-node index of each residence (0..M-1)
    Rn        (M,M)  LinDistFlow matrix restricted to the constraint nodes
    group            torch.distributed process group when residences are sharded
    comm_hook        optional `fn(host_array, op)`: the all-reduce of the node sums over the caller's
                     own transport instead of RCCL (revs_comm_create_hook; op 0 sum, 2 max, 3 min,
                     in place on a float64 numpy view).  Default with a `group` whose backend is
                     not nccl (gloo: ranks sharing one device, CPU-side transports): all_reduce
                     over that group.
    feeder           optional (parent, edge_r, cons_of): the radial feeder behind Rn as a tree --
                     parent[i] (-1: hangs off the substation), resistance of the edge to the
                     parent, constraint row of tree node i (or -1).  With it the steady state
                     evaluates R p in O(nodes) (OperatorOptions.voltage); the constructor checks
                     that the tree reproduces Rn.
    """

    # the HIP stream of a run in progress: step / run_steps / run look torch's current stream up once at their entry
    # (2.5 us a look-up, ~3 per iteration otherwise) and every launch inside goes there
    _stream_pin = None


    def __init__(self, cost, homes, load, node_of, Rn, kappa=5.0, vset=1.0, vlow=0.95,
                 vhigh=1.05, mode="binary", device="cuda:0", pdhg=None,
                 op: OperatorOptions | None = None, group=None, node_counts=None,
                 pdhg_warm=True, feeder=None, comm_hook=None, _kernels=None):
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
        # (a comm.LocalRanks rank: logical ranks inside one process -- a host-side transport like a gloo group)
        self._local_group = group is not None and hasattr(group, "allreduce_host")
        self._host_group = (group is not None and _kernels is None
                            and (self._local_group or torch.distributed.get_backend(group) != "nccl"))
        if self._local_group:
            counts = group.allreduce_host(counts.astype(np.float64), 0).astype(np.int64)
        elif group is not None:
            ct = torch.from_numpy(counts) if self._host_group else torch.from_numpy(counts).to(self.dev)
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
        self.nks = int(self.op.newton_nks) if self.op.newton_nks else int(min(16, max(1, M // 128)))       # column slabs of the model Hessian (121144, M = 1126: 8; 4 / 16 / 32 measured slower, r05)
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
        self._eval_seq = 0.0                           # tags of the other evaluations: 1, 2, ...
        self._pending_tag = [None, None]
        self.chain_hist = [0, 0]                       # chained Newton iterations kept / redone
        self._y_spare = None                           # third multiplier array (folded chain)
        self.fold_steps = 0                            # folded-chain iterations that went on from their own Newton step
        self._fold_resume = False                      # the folded chain's pipeline is primed for the next iteration
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
        # ... and the word check_status() has the residences' status bits OR-ed into (revs_status_or)
        self._flag_h = torch.zeros(2, dtype=torch.int32, pin_memory=cuda)
        self._flag_np = self._flag_h.numpy()
        self._flag_dev = None
        # ... and the residual records (revs_residual_finalize's float[4]) of iterations outside the streaming loop,
        # which run(eps=) collects until its stopping rule is next evaluated
        self._dmx_h = torch.zeros(64, 4, dtype=torch.float32, pin_memory=cuda)
        self._dmx_np = self._dmx_h.numpy()
        self._dmx_dev = None
        if cuda:
            for name, t in (("alpha_dev", self.alpha_h), ("info_dev", self.info_h), ("_flag_dev", self._flag_h),
                            ("_dmx_dev", self._dmx_h)):
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
        if cuda and self.op.solver == "newton" and _kernels is None and self.op.native_plan:
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
            # round 2; OperatorOptions(recompute_pe_new=False) restores the read)
            d.recompute_pe_new = int(bool(self.op.recompute_pe_new))
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
            check(self.lib.revs_plan_set_fold_redo(self._plan, int(self.op.fold_redo)), "revs_plan_set_fold_redo")
            check(self.lib.revs_plan_set_kadd_cold(self._plan, int(self.op.newton_kadd_cold), int(self.op.newton_kadd_cold_at)),
                  "revs_plan_set_kadd_cold")
            no = _lib.NewtonOpts(ptr(self.k_slabs), self.nks, self.alpha_h.data_ptr(), self.alpha_dev,
                                 self.info_h.data_ptr(), int(self.op.newton_max), int(self.op.newton_ls))
            check(self.lib.revs_plan_set_newton(self._plan, C.byref(no)), "revs_plan_set_newton")
        # third node-sum buffer and the feeder as a tree: streaming steady state
        self.p_alt2 = nz()
        self._burst = max(1, int(self.op.stream_burst))
        self._p_clear = None             # the node-sum array the last streaming launch cleared
        self._tree = None
        self._tree_newton = False        # the fused launches of the chained iteration hold the tree (256 x 8 positions)
        self._tree_eval = False          # Newton evaluations judge their rows by the tree form (every shape)
        self._comm = None
        recovered = False
        if (feeder is None and self.op.voltage in ("auto", "tree") and M <= 4096 and self.op.solver == "newton"
                and _kernels is None):
            # the caller holds only the matrix (lpsolver.py:184-189 hands the solver R, not the network): a radial
            # feeder is recovered from it -- junctions without a residence become extra tree nodes -- and verified
            # against Rn below like a feeder the caller passed; anything else stays on the dense product
            feeder = tree_from_R(Rn)
            recovered = feeder is not None
        if feeder is not None and self.op.voltage in ("auto", "tree"):
            par, er, cons = feeder
            tr = None
            if len(par) <= _lib.TREE_MAX:          # (padded to a multiple of 8 below)
                try:
                    tr = feeder_tree(par, er, cons, counts > 0)
                except ValueError:                 # (row indices beyond the packed fields)
                    if self.op.voltage == "tree":
                        raise
            rejected = False
            if tr is not None:
                # verified row by row: every constrained row's voltage for two random injection profiles, relative to
                # THAT row's own magnitude (a row whose entries are small against the feeder's largest is checked as
                # tightly as the largest)
                probe = np.random.default_rng(0).uniform(0.5, 1.5, (M, 2)) * (counts > 0)[:, None]
                ref = (Rn @ probe) * (counts > 0)[:, None]
                got = tree_voltage_host(tr, probe)
                row_scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-300)
                # (1e-13 of the largest voltage: the prefix sums' own rounding, which a small row cannot be held to)
                if (np.abs(got - ref) > np.maximum(1e-9 * row_scale, 1e-13 * np.abs(ref).max())).any():
                    if not recovered:
                        raise ValueError("feeder: the tree does not reproduce Rn (R[i][j] = 2 x the "
                                         "resistance shared by the substation->i and ->j paths)")
                    tr, rejected = None, True              # (Rn is not a radial feeder's matrix: dense product)
            if tr is not None:
                self._tree_host = tr
                self._tree_dev = {k: up(tr[k].view(np.int64) if k == "pack" else tr[k]) for k in ("pack", "w")}
                self._tree = _lib.Tree(tr["n"], ptr(self._tree_dev["pack"]), ptr(self._tree_dev["w"]))
                if self._plan is not None:
                    check(self.lib.revs_plan_set_tree(self._plan, C.byref(self._tree)),
                          "revs_plan_set_tree")
                    self._tree_newton = tr["n"] <= _lib.TREE_SWEEP_MAX
                    self._tree_eval = True
            elif self.op.voltage == "tree" and rejected:
                raise ValueError('OperatorOptions(voltage="tree"): Rn is not the matrix of a radial feeder (the tree recovered '
                                 "from it does not reproduce its rows); pass feeder= or use the dense product")
            elif self.op.voltage == "tree":
                raise ValueError(f"feeder has {len(par)} nodes; the tree form holds {_lib.TREE_MAX}")
        elif self.op.voltage == "tree":
            raise ValueError('OperatorOptions(voltage="tree") needs feeder=')
        self.tree_recovered = bool(recovered and self._tree is not None)   # the tree form runs on a tree recovered from Rn
        self._hook_ref = None
        if group is not None and cuda and _kernels is None and (comm_hook is not None or self._host_group):
            # the caller's transport behind the library's communicator (revs_comm_create_hook)
            if self._local_group:
                ws, rk = group.size, group.rank_id
            else:
                ws, rk = torch.distributed.get_world_size(group), torch.distributed.get_rank(group)
            from .comm import group_allreduce_hook, host_hook
            self._hook_ref = host_hook(comm_hook if comm_hook is not None else
                                       (group.allreduce_host if self._local_group else group_allreduce_hook(group)))
            self._comm = self.lib.revs_comm_create_hook(self._hook_ref, None, rk, ws)
            if not self._comm:
                raise _lib.RevsError("revs_comm_create_hook failed: " + self.lib.revs_last_error().decode())
            if self._plan is not None:
                check(self.lib.revs_plan_set_comm(self._plan, self._comm), "revs_plan_set_comm")
        elif group is not None and cuda and _kernels is None and self.op.library_comm:
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
        self._block = 0
        self._sets = None               # buffer pools of the block form (allocated on first use)
        # max_h diff[h] (lpsolver.py:284) of the iterations whose sweeps folded it on the device:
        # {iteration number (1-based, as the reference's diff[k]): value}
        self.max_diff = {}               # (property: also clears the bursts' records not yet folded in)
        self._dmax_buf = (C.c_double * 1024)()
        self._dmax_addr = C.addressof(self._dmax_buf)
        self.stream_calls = []          # (launches enqueued, iterations kept) of every native burst
        # arguments of revs_plan_stream_run, built once (a burst of 20 sweeps is 0.4 ms: every
        # microsecond of Python around it is a microsecond of idle GPU)
        self._stream_st = _lib.StreamState()
        self._stream_st_ref = C.byref(self._stream_st)
        self._stream_out = (C.c_int32(), C.c_double())
        self._stream_out_ref = (C.addressof(self._stream_out[0]), C.addressof(self._stream_out[1]))
        self._scale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        self._pn0 = None
        if (self._plan is not None and self._tree is not None and self.op.stream_block > 1
                and self.recompute_pe_new and (self._comm is not None or self.op.stream_block_single)):
            self._block = min(int(self.op.stream_block), _lib.STREAM_BLOCK_MAX)
            check(self.lib.revs_plan_set_stream_block(self._plan, self._block, int(self.op.stream_overlap)),
                  "revs_plan_set_stream_block")
            self._inner = max(1, min(int(self.op.stream_inner), int(self.lib.revs_agent_max_inner(T, int(self.pdhg.lanes)))))
            check(self.lib.revs_plan_set_stream_inner(self._plan, self._inner), "revs_plan_set_stream_inner")
            self._sets_st = _lib.StreamSets()
            self._sets_by, self._sets_sig, self._pn_ptr = {}, None, {}
            self._sets_ref = C.byref(self._sets_st)
        # R (float) for the voltage check
        self.R32 = up(Rn.astype(np.float32))
        self.node_load = torch.zeros(M, T, **f32)
        self.volt = torch.zeros(M, T, **f32)
        # What the run loops would allocate on first use, now (OperatorOptions.preallocate): the pools the state
        # rotates through, the folded chain's third multiplier array, the plan's ring / events / chain buffers
        # (revs_plan_prepare) -- a fresh engine's first run then makes no allocation between its launches
        # (~0.5 ms of the 2.2 ms transient at 100 000 x 24, tools/transient_hostgaps.py).
        if self._plan is not None and self._tree is not None and self.op.preallocate:
            self._state_pools()
            self._y_spare = torch.zeros_like(self.yd[0])
            check(self.lib.revs_plan_prepare(self._plan), "revs_plan_prepare")
            if cuda:
                torch.cuda.synchronize(self.dev)

    # ------------------------------------------------------------------ util
    # max_h diff[h] per iteration (1-based) -- a dict; the streaming bursts leave their records as (first iteration, values)
    # and the dict is brought up to date when somebody looks (a burst of 20 iterations is 150 us: the 3 us of twenty dict
    # entries per burst belong to whoever reads them)
    _max_diff = None
    _max_diff_bursts = ()

    @property
    def max_diff(self):
        if self._max_diff_bursts:
            md = self._max_diff
            for it, vals in self._max_diff_bursts:
                md.update(zip(range(it + 1, it + 1 + len(vals)), vals))
            self._max_diff_bursts = []
        return self._max_diff

    @max_diff.setter
    def max_diff(self, value):
        self._max_diff = value
        self._max_diff_bursts = []

    @contextlib.contextmanager
    def _capture(self, g):
        """torch.cuda.graph(g) for the engine's own launches: inside it `stream` is torch's capture stream, not the
        one a run in progress has pinned."""
        pin, self._stream_pin = self._stream_pin, None
        try:
            with torch.cuda.graph(g):
                yield
        finally:
            self._stream_pin = pin

    @property
    def stream(self):
        if self._stream_pin is not None:
            return self._stream_pin
        if self.dev.type != "cuda":
            return None
        return torch.cuda.current_stream(self.dev).cuda_stream

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
        if getattr(self, "_local_group", False):                  # logical ranks in this process: through the host, as doubles
            code = 0 if op in (None, RO.SUM) else (2 if op == RO.MAX else 3)
            h = t.detach().cpu().to(torch.float64).contiguous()
            self.group.allreduce_host(h.numpy().reshape(-1), code)
            t.copy_(h.to(t.dtype))
            return
        if getattr(self, "_host_group", False) and t.is_cuda:     # a CPU-side group: through the host
            h = t.cpu()
            torch.distributed.all_reduce(h, op=op or RO.SUM, group=self.group)
            t.copy_(h)
            return
        torch.distributed.all_reduce(t, op=op or RO.SUM, group=self.group)

    def _gemm1(self, At, B, Cslabs):
        check(self.lib.revs_gemm_tn_f64_split(self.M, self.T, self.M, ptr(At), ptr(B),
                                              ptr(Cslabs), self.ksplit1, self.stream),
              "revs_gemm_tn_f64_split")

    # -------------------------------------------------------------- operator
    # (dual Newton path: operator_newton.py; ADMM forms: operator_admm.py; steady state:
    # steady_state.py -- mixed into this class)
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
        """Utility(graph, P_est[k], P_sch[k], G[k]).solve() -> P_est[k+1]
        (lpsolver.py:256-259), written to self.P_est_new.  `admm_only`: skip the dual
        Newton attempt (the caller has just seen it fail for this state)."""
        self._p_clear = None
        o, lib, M, T, st = self.op, self.lib, self.M, self.T, self.stream
        self._fused_ready = False
        self._fold_resume = False
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
        self._fold_resume = False
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

    @_on_current_stream
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
            self._fold_resume = False
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
            if not events and self._fold_ok():
                self._chain_run(1, write_sc)         # (books the iteration itself)
                return
            self._fold_resume = False
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

    @_on_current_stream
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
            if ((native or (o.solver == "newton" and self._fold_ok())) and o.chain and self._chain_ok
                    and not (o.speculate and self._spec_ok)):
                done += self._chain_run(count - done)        # (sharded: the folded chain only)
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

    def check_status(self, launch_only=False, deferred=False):
        """Raise if a residence reported 'no solution' (lpsolver.py:153-155) or a PDHG residence
        stopped at its iteration cap, in any sweep since the last check -- a collective decision
        when residences are sharded (every rank raises, or none).  Call at a point where the
        stream has been synchronised (run(), result() and residuals() do).
        launch_only / deferred (one GPU, no group): the reduction of the status words is enqueued now and read by a
        second call with deferred=True -- run() puts the next iteration between the two, so that the report that
        follows the first iteration (as the reference's does) costs the host no wait for that iteration's sweep."""
        if deferred and not (self.n and self._flag_dev is not None and self.group is None):
            return
        if launch_only and not (self.n and self._flag_dev is not None and self.group is None):
            launch_only = False
        if self._plan is not None and not deferred:
            f = int(self.lib.revs_plan_status_flags(self._plan, 1))
        else:
            f = 0
        if self.n and self._flag_dev is not None:
            # bits 0-2 of the last sweep's per-residence status words, OR-ed on the device into a pinned word: one
            # small launch and a stream synchronise (as six torch reductions and a read-back this check was 0.3 ms
            # of host time in front of the second iteration of every run -- tools/transient_hostgaps.py, r05)
            if not deferred:
                self._flag_np[0] = 0
                check(self.lib.revs_status_or(self.n, ptr(self.status), self._flag_dev, self.stream), "revs_status_or")
                if launch_only:          # (read by check_status(deferred=True) behind the next iteration's own wait)
                    self._flag_np[1] = f
                    return
            else:
                f |= int(self._flag_np[1])
            torch.cuda.current_stream(self.dev).synchronize()
            f |= int(self._flag_np[0]) & 7
        elif self.n:    # (host stand-in of the kernels: one read-back)
            st = self.status
            f |= sum(int(v) for v in torch.stack([(st & b).max() for b in (1, 2, 4)]).cpu().tolist())
        if self.group is not None:
            t = torch.tensor([float(f & 1), float((f >> 1) & 1), float((f >> 2) & 1)], dtype=torch.float64, device=self.dev)
            self._allreduce(t, torch.distributed.ReduceOp.MAX)
            a, b, c = t.cpu().tolist()
            f = int(a) | (int(b) << 1) | (int(c) << 2)
        if f & 4:
            # bit 2: a PDHG residence's KKT polish did not settle within its six steps -- its schedule is PDHG's
            # iterate at the loosened 1e-4 step tolerance (feasible, a little off the optimum): counted, warned once
            self.polish_unsettled = getattr(self, "polish_unsettled", 0) + 1
            if self.polish_unsettled == 1:
                import warnings
                warnings.warn("revs_admm_amd: a residence's KKT polish did not settle (status bit 2); its schedule is "
                              "PDHG's iterate at the 1e-4 step tolerance", RuntimeWarning, stacklevel=2)
        if f & 1:
            raise _lib.RevsError("No solution found (lpsolver.py:153-155): a residence's "
                                 "charging window cannot reach 90% state of charge")
        if f & 2:
            raise _lib.RevsError("REVS_ENOTCONV: a residence's PDHG iteration reached max_iter "
                                 f"({self.pdhg.max_iter}) before its tolerance "
                                 f"({self.pdhg.tol:g}{' = automatic' if self.pdhg.tol == 0 else ''})")

    def _resolve_max_diff(self, pending):
        """Fetch the device-side maxima collected for iterations outside the streaming loop (one transfer), over every
        rank's residences when they are sharded (see _max_diff_all_ranks), into self.max_diff; empties `pending`."""
        if pending and isinstance(pending[0][1], int):        # records in pinned memory (one GPU): wait, read
            torch.cuda.current_stream(self.dev).synchronize()
            for it, slot in pending:
                self.max_diff[it] = float(self._dmx_np[slot, 2])
            pending.clear()
            return
        mx = torch.stack([v for _, v in pending]).to(torch.float64)
        if self.group is not None:
            self._allreduce(mx, torch.distributed.ReduceOp.MAX)
        for (it, _), v in zip(pending, mx.cpu().tolist()):
            self.max_diff[it] = float(v)
        pending.clear()

    def _max_diff_all_ranks(self):
        """max_h diff[h] of the iteration just finished over EVERY rank's residences (the streaming
        loop's records are global already: each rank's partial maxima travel with the all-reduce).
        run(eps=) decides on it when to stop, and a rank that stopped on its own residences' maximum
        alone would leave the others waiting in their next all-reduce."""
        mx = torch.tensor([float(self.diff.max().item()) if self.n else 0.0], dtype=torch.float64,
                          device=self.dev)
        if self.group is not None:
            self._allreduce(mx, torch.distributed.ReduceOp.MAX)
        return float(mx.item())

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

    @_on_current_stream
    def run(self, iter_max=15, eps=None, patience=8, history=True):
        """Full solve_ADMM loop; returns diff (iterations, n) in the caller's home order.
        The per-iteration diff of every residence (lpsolver.py:284) is collected on the device --
        the steady-state launches write their row themselves -- and fetched in pieces of at most
        2 GB: the host is not in the loop of the iterations, only a residence that cannot reach
        90 % SOC is reported right after the first iteration, as the reference does.
        The reference always runs iter_max iterations (lpsolver.py:254).  With `eps` the run also
        stops once max_h diff[h] <= eps has held for `patience` consecutive iterations (diff[2] is
        ~0 in every run and rows that start to bind make it jump: one iteration proves nothing) --
        judged on the records the streaming launches leave (`max_diff`: folded on the device, no
        read-back of diff), at the end of the burst in which it happened; one more iteration then
        writes the schedules.  `converged_at`: the first iteration of that stretch, or None.
        `history=False`: the per-residence diff of every iteration is neither kept nor fetched (at
        100 000 residences x 500 iterations it is 200 MB to carry to the host, more time than the
        iterations themselves take); returns the number of iterations run, `max_diff` holds max_h diff[h]
        of the streamed ones."""
        rows = max(1, min(iter_max, int(2e9) // (4 * max(self.n, 1)))) if history else iter_max
        if history:
            diffs = torch.empty((iter_max, self.n), dtype=torch.float32)
            hist = torch.empty((rows, self.n), dtype=torch.float32, device=self.dev)
            inv = torch.from_numpy(np.ascontiguousarray(self.inv_perm, dtype=np.int64)).to(self.dev)
        self.converged_at = None
        k, it0, good, seen, stop = 0, self.iteration, 0, self.iteration, False
        cap, pending, status_pending = 64, [], False
        while k < iter_max:
            base, r = k, 0
            while r < rows and k < iter_max:
                last = k == iter_max - 1 or stop
                if (not last and self._stream_ok() and self._fused_ready):
                    # (with eps: bursts of at most 64, so that the run ends soon after the stretch.  max_h diff is no
                    # geometric decay one could size the bursts by: on the bench workload it sits on a plateau of ~6e-4 from
                    # iteration 100 to 480 and falls to 7e-5 within ten iterations -- tools/newton_trace.py)
                    done = self._stream_run(min(rows - r, iter_max - 1 - k, cap if eps is not None else rows),
                                            hist[r:] if history else None)
                else:
                    self.step(write_sc=last)
                    if history:
                        hist[r].copy_(self.diff)
                    done = 1
                    if k == 0:
                        # the reference reports a residence without a solution right after the first iteration
                        # (lpsolver.py:153-155): the reduction is enqueued here and read behind the next iteration
                        # (sharded: the whole check now, which synchronises once)
                        self.check_status(launch_only=True)
                        status_pending = True
                    if eps is not None and self.iteration not in self.max_diff:
                        # (max_h diff of an iteration outside the streaming loop: reduced on the device now, fetched with
                        # the others' when the stopping rule is next evaluated -- no host wait per iteration)
                        if self._dmx_dev is not None and self.group is None and self.n:
                            # (one GPU: the residual kernels write their record into a slot of pinned memory -- no
                            # torch reduction per iteration, no stack-and-copy when the records are wanted)
                            slot = len(pending)
                            check(self.lib.revs_residual_finalize(ptr(self.diff), ptr(self.dsq), self.n, self.T, self.kappa,
                                                                  eps, ptr(self.res_scratch), self._dmx_dev + 16 * slot,
                                                                  self.stream), "revs_residual_finalize")
                            pending.append((self.iteration, slot))
                        else:
                            pending.append((self.iteration, self.diff.max() if self.n else torch.zeros((), dtype=torch.float32, device=self.dev)))
                k += done
                r += done
                if status_pending and k >= 2:
                    status_pending = False
                    self.check_status(deferred=True)
                if stop:
                    break
                if eps is not None and pending and (done > 1 or len(pending) >= min(patience, 64) or k >= iter_max - 1):
                    self._resolve_max_diff(pending)
                if eps is not None:                  # the stretch of iterations at or below eps so far
                    while seen < it0 + k - len(pending) and not stop:
                        seen += 1
                        good = good + 1 if self.max_diff.get(seen, np.inf) <= eps else 0
                        if good >= patience:
                            self.converged_at = seen - patience + 1
                            stop = True
            # (back to the caller's home order on the device: one gather, one copy)
            if history:
                diffs[base:k].copy_(hist[:r].index_select(1, inv))
            if stop and last:
                break
        if status_pending:
            self.check_status(deferred=True)
        self.check_status()
        return diffs[:k].numpy() if history else k

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
        self._fold_resume = False
        if iteration is not None:
            self.iteration = int(iteration)

    def reset(self):
        """Back to iteration 0 of lpsolver.py:244-246 (P_est = P_sch = G = 0, no multipliers, cold operator):
        the same engine -- buffers, plan, loaded code objects -- for another run of the same problem."""
        z = np.zeros((self.n, self.T), np.float32)
        self.set_state(z, z, z, iteration=0)
        for y in self.yd:
            y.zero_()
        if self.pdhg_dual is not None:
            self.pdhg_dual.zero_()
        self._y_support = self._spec_ok = self._chain_few = False
        self._sup = None
        self._spec_wait, self._spec_back = 0, 1
        self.op_cold = self._fast_cold = True
        self._burst = max(1, int(self.op.stream_burst))
        self.max_diff = {}               # (property: also clears the bursts' records not yet folded in)
        for h in (self.op_iters_hist, self.op_path_hist, self.newton_hist, self.stream_calls):
            h.clear()
        self.spec_hist, self.chain_hist, self.model_calls, self.fold_steps = [0, 0], [0, 0], [0, 0], 0

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
