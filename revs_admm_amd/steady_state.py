"""The steady state of lpsolver.solve_ADMM's loop (reference lpsolver.py:254-287) while the
operator's multipliers are zero: bursts enqueued by the native loops -- revs_plan_stream_run_blocks
(the default since round 3: eight iterations per launch, verdicts by blocks, four rotating sets of
state buffers) or revs_plan_stream_run (every launch judges itself, section
3.5) -- verdicts and the convergence record on the device; and what follows a failed verdict.
Methods of AdmmEngine (mixed in by engine.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr


def _dp(t) -> int:
    """data_ptr() of a long-lived buffer, remembered on the tensor object itself (the bursts of
    the streaming loop are short: a dozen data_ptr() calls are microseconds of an idle GPU)."""
    d = getattr(t, "_revs_dp", None)
    if d is None:
        d = t.data_ptr()
        try:
            t._revs_dp = d
        except AttributeError:
            pass
    return d


class SteadyStateMixin:
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

    def _stream_ok(self):
        """The steady state as one launch per iteration (revs_plan_stream_run) applies: a plan
        with the feeder's tree, no multipliers, speculation allowed, and -- sharded -- the
        library's own communicator."""
        o = self.op
        return (self._plan is not None and self._tree is not None and o.solver == "newton"
                and (self._block or self._tree.n <= _lib.TREE_SWEEP_MAX)     # (big trees: judged by blocks only)
                and o.speculate and self._spec_ok and o.fuse_home_pass and not self._y_support
                and (self.group is None or self._comm is not None))

    def _stream_run(self, count, hist=None):
        """Up to `count` steady-state iterations (at most the current burst), one launch each,
        enqueued in one go by the native loop; the first iteration whose rows are not within
        tolerance silences the launches behind it and is finished here as step() finishes a
        discarded speculative sweep.  Returns the number of iterations done (at least one).
        `hist`: float[>= count][n] on the device -- row i receives the residences' diff of the
        i-th iteration done here (run(): solve_ADMM's per-iteration diff without a host round
        trip per iteration)."""
        o = self.op
        self._fold_resume = False
        if self._block:
            return self._stream_run_blocks(count, hist)
        p0 = self._fused_p
        d0 = _dp(p0)
        if self._pn0 is None:
            self._pn0 = self.pnq[0]
        rest = [b for b in (self._pn0, self.p_alt, self.p_alt2) if _dp(b) != d0]
        # p[1] must be zero on entry.  The last launch of a fully kept call has just cleared the
        # array that is p[1] now (roles rotate): no fill launch then
        if self._p_clear is not None and _dp(self._p_clear) != d0:
            dc = _dp(self._p_clear)
            rest = [self._p_clear] + [b for b in rest if _dp(b) != dc]
        else:
            rest[0].zero_()
        self._p_clear = None
        if self.group is not None and not self._ar_ahead:
            self._allreduce(p0)
        pes = (self.P_est, self.P_est_new, self.P_est_alt)
        pss, gs, ps = (self.P_sch, self.P_sch_alt), (self.G, self.G_alt), (p0, rest[0], rest[1])
        st = self._stream_st
        by = {}
        for i in range(3):
            a, b = _dp(pes[i]), _dp(ps[i])
            st.p_est[i], st.p[i] = a, b
            by[a], by[b] = pes[i], ps[i]
        for i in range(2):
            a, b = _dp(pss[i]), _dp(gs[i])
            st.p_sch[i], st.gamma[i] = a, b
            by[a], by[b] = pss[i], gs[i]
        kept, rm = self._stream_out
        count = min(count, self._burst)
        st.diff_hist = None if hist is None else hist.data_ptr()
        check(self.lib.revs_plan_stream_run(self._plan, count, self._stream_st_ref, self._scale, o.eps,
                                            self._stream_out_ref[0], self._stream_out_ref[1], self.stream),
              "revs_plan_stream_run")
        n = kept.value
        self.stream_calls.append((count, n))
        self._burst = min(4 * self._burst, o.stream_burst_max) if n == count else o.stream_burst
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
            if hist is not None:
                self.diff.copy_(hist[n - 1])         # (self.diff: always the last iteration's)
            return n
        # iteration n's verdict failed (its sweep wrote to the spares only; every launch behind
        # it was a no-op): finish it as step() does for a discarded speculative sweep
        self._fused_ready = False
        self._spec_discard(None, False)
        self.P_est, self.P_est_new = self.P_est_new, self.P_est
        self.iteration += 1
        if hist is not None:
            hist[n].copy_(self.diff)
        return n + 1

    def _state_pools(self):
        """The pools the residences' state buffers rotate through (allocated on first use): four sets for the
        block form's roll-back without copies, of which the folded chain borrows a third set for the sweep it
        enqueues unjudged.  self.P_est / P_est_new / P_est_alt / P_sch / P_sch_alt / G / G_alt are always members."""
        if self._sets is None:
            z = lambda t: torch.zeros_like(t)
            self._sets = dict(pe=[self.P_est, self.P_est_new, self.P_est_alt, z(self.P_est), z(self.P_est)],
                              ps=[self.P_sch, self.P_sch_alt, z(self.P_sch), z(self.P_sch)],
                              g=[self.G, self.G_alt, z(self.G), z(self.G)],
                              y=([self.pdhg_dual] + [z(self.pdhg_dual) for _ in range(3)]
                                 if self.pdhg_dual is not None else None))
        return self._sets

    def _stream_run_blocks(self, count, hist=None):
        """_stream_run with the verdicts taken by blocks and `stream_inner` iterations per launch
        (revs_plan_stream_run_blocks): the state rotates through four sets of buffers, so that a
        failed verdict deep inside a burst is undone without copies."""
        o = self.op
        self._state_pools()
        p0 = self._fused_p
        if self.group is not None and not self._ar_ahead:
            self._allreduce(p0)
        pool = self._sets
        st, by = self._sets_st, self._sets_by
        # The native call leaves the sets in the order the next call wants them (slot 0 = the state).  As long
        # as nothing else has touched the engine's tensors since, the struct is handed back as it is.
        sig = (id(self.P_est), id(self.P_est_new), id(self.P_sch), id(self.G), id(self.pdhg_dual))
        if sig != self._sets_sig:
            def roles(first, tensors, skip=()):
                rest = [t for t in tensors if t is not first and all(t is not x for x in skip)]
                return [first] + rest

            pes = roles(self.P_est, pool["pe"], skip=(self.P_est_new,))
            pss, gs = roles(self.P_sch, pool["ps"]), roles(self.G, pool["g"])
            ys = roles(self.pdhg_dual, pool["y"]) if pool["y"] is not None else [None] * 4
            assert len(pes) == len(pss) == len(gs) == len(ys) == 4
            for i in range(4):
                for arr, t in ((st.p_est, pes[i]), (st.p_sch, pss[i]), (st.gamma, gs[i]), (st.pdhg_dual, ys[i])):
                    arr[i] = None if t is None else _dp(t)
                    if t is not None:
                        by[_dp(t)] = t
            st.p_est_next = _dp(self.P_est_new)
        if self._pn0 is None:
            self._pn0 = self.pnq[0]
            self._pn_ptr = {id(self._pn0): _dp(self._pn0), id(self.p_alt): _dp(self.p_alt)}
        # (the sums handed over for the next call; by address: self.pnq[0] is a fresh view object wherever it is taken)
        p0_out = self.p_alt if _dp(p0) == _dp(self._pn0) else self._pn0
        pp = self._pn_ptr
        st.p0 = pp[id(p0)] if id(p0) in pp else _dp(p0)
        st.p0_out = pp[id(p0_out)]
        st.diff_hist = None if hist is None else hist.data_ptr()
        count = min(count, self._burst, 1000)
        kept, rm = self._stream_out
        check(self.lib.revs_plan_stream_run_blocks(self._plan, count, self._sets_ref, self._scale, o.eps,
                                                   self._stream_out_ref[0], self._stream_out_ref[1],
                                                   self._dmax_addr, self.stream),
              "revs_plan_stream_run_blocks")
        n = kept.value
        self.stream_calls.append((count, n))
        self._burst = min(4 * self._burst, o.stream_burst_max) if n == count else o.stream_burst
        self.P_est, self.P_est_alt = by[st.p_est[0]], by[st.p_est[1]]
        self.P_sch, self.P_sch_alt = by[st.p_sch[0]], by[st.p_sch[1]]
        self.G, self.G_alt = by[st.gamma[0]], by[st.gamma[1]]
        if pool["y"] is not None:
            self.pdhg_dual = by[st.pdhg_dual[0]]       # (the plan already points at it)
        self._sets_sig = ((id(self.P_est), id(self.P_est_new), id(self.P_sch), id(self.G), id(self.pdhg_dual))
                          if n == count else None)
        self._p_clear = None
        self._prod_ahead = False
        self._ar_ahead = self.group is not None
        if n == count:
            self._fused_p = p0_out
        it = self.iteration
        if n:
            self._max_diff_bursts.append((it, self._dmax_buf[:n]))      # (folded into max_diff when it is next read)
        if n:
            self.op_iters_hist.extend([1] * n)
            self.op_path_hist.extend(["dual"] * n)
            self.newton_hist.extend([(0, 1, 0)] * n)
            self.op_converged = True
            self.spec_hist[0] += n
            self._spec_back = 1
            self.iteration += n
        if n == count:
            if hist is not None:
                self.diff.copy_(hist[n - 1])
            return n
        self._fused_ready = False
        self._spec_discard(None, False)
        self.P_est, self.P_est_new = self.P_est_new, self.P_est
        self.iteration += 1
        if hist is not None:
            hist[n].copy_(self.diff)
        return n + 1
