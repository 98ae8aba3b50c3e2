"""The steady state of lpsolver.solve_ADMM's loop (reference lpsolver.py:254-287) while the
operator's multipliers are zero: one launch per iteration, verdicts on the device, bursts enqueued
by the native loop revs_plan_stream_run; and what follows a failed verdict.
Methods of AdmmEngine (mixed in by engine.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr


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
        # (verdicts by blocks: the sweeps accumulate into the plan's ring, p[1] and p[2] are not used)
        if self._block:
            pass
        elif self._p_clear is not None and self._p_clear.data_ptr() != p0.data_ptr():
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
        self.stream_calls.append((count, n))
        self._burst = min(4 * self._burst, o.stream_burst_max) if n == count else o.stream_burst
        by = {t.data_ptr(): t for t in pes + pss + gs + ps}
        self.P_est, self.P_est_new, self.P_est_alt = (by[st.p_est[i]] for i in range(3))
        self.P_sch, self.P_sch_alt = by[st.p_sch[0]], by[st.p_sch[1]]
        self.G, self.G_alt = by[st.gamma[0]], by[st.gamma[1]]
        self._fused_p = by[st.p[0]]
        self._p_clear = by[st.p[1]] if (n == count and not self._block) else None
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
