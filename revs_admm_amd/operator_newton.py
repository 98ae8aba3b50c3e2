"""The operator QP (reference lpsolver.py:163-238) through its dual: semismooth Newton on the
voltage-row multipliers -- evaluations, candidate models, Armijo steps on
the host, and the binding steady state enqueued whole (one Newton iteration per ADMM iteration).
Methods of AdmmEngine (mixed in by engine.py); kernels: csrc/newton_kernels.hip, gemm_kernels.hip."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr


class DualNewtonMixin:
    def _dual_phase(self, phase: int, y, use_y: bool, k: int, kadd=None, dense=False):
        lib, M, T = self.lib, self.M, self.T
        kadd = self.op.newton_kadd if kadd is None else int(kadd)
        # an evaluation that ends with its own selection tags the stats block (pinned host
        # memory, written by the selection's workgroups behind a system-scope fence) with a
        # fresh number: _dual_wait polls for it -- no event record on the stream, no
        # synchronize on the host
        # (n + 0.5: the native loops tag the same blocks with whole numbers, +/-)
        tag = 0.0
        if (phase & 2) and not (phase & 4):
            self._eval_seq += 1.0
            tag = self._pending_tag[k] = self._eval_seq + 0.5
        if self._tree_eval and self._tree is not None and not dense:
            # the feeder as a tree: R p and the rows of every slot in O(nodes), one workgroup per slot
            check(lib.revs_op_dual_evaluate_tree(
                phase, M, T, ptr(self.node_ptr), ptr(self.P_est), ptr(self.P_sch), ptr(self.G),
                ptr(self.R64), C.byref(self._tree), ptr(y), int(use_y), self.kappa, self.vlo, self.vhi,
                kadd, self.ksplit1, ptr(self.d_sl), ptr(self.pnq), ptr(self.P_est_new),
                ptr(self.vfull), ptr(self.violw), ptr(self.d_part), ptr(self.c_idx[k]),
                ptr(self.c_cnt[k]), ptr(self.c_val[k]), self.stats_dev[k], tag, self.stream),
                "revs_op_dual_evaluate_tree")
            return
        check(lib.revs_op_dual_evaluate(
            phase, M, T, ptr(self.node_ptr), ptr(self.P_est), ptr(self.P_sch), ptr(self.G),
            ptr(self.R64), ptr(self.R64T), ptr(y), int(use_y), self.kappa, self.vlo, self.vhi,
            kadd, self.ksplit1, ptr(self.d_sl), ptr(self.v_sl), ptr(self.pnq),
            ptr(self.P_est_new), ptr(self.vfull), ptr(self.violw), ptr(self.d_part),
            ptr(self.c_idx[k]), ptr(self.c_cnt[k]), ptr(self.c_val[k]), self.stats_dev[k],
            tag, ptr(self.tile_cnt), self.stream), "revs_op_dual_evaluate")

    def _dual_home_pass_rows(self, y, sup: int):
        """Phase 1 of an evaluation with d = R^T y / kappa taken from the few rows listed in
        candidate set `sup` (they include every row with y != 0) instead of a dense product."""
        check(self.lib.revs_op_dual_eval_rows(
            self.M, self.T, ptr(self.node_ptr), ptr(self.P_est), ptr(self.P_sch), ptr(self.G),
            ptr(self.R64), ptr(self.c_idx[sup]), ptr(self.c_cnt[sup]), ptr(y), self.kappa,
            ptr(self.pnq), ptr(self.P_est_new), self.stream), "revs_op_dual_eval_rows")

    def _dual_launch(self, y, use_y: bool, k: int, full: bool = True, sup=None, record=True, kadd=None, dense=False):
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
            self._dual_phase(2, y, use_y, k, kadd, dense)
        elif self.group is None:
            self._dual_phase(3, y, use_y, k, kadd, dense)
        else:
            self._dual_phase(1, y, use_y, k, kadd, dense)
            self._allreduce(self.pnq if full else self.pnq[0])   # the only exchange
            self._dual_phase(2, y, use_y, k, kadd, dense)

    def _dual_complete(self, y, use_y: bool, k: int):
        """After a full=False evaluation that did not settle the solve: exchange N and the
        dual value too and redo the row bookkeeping (the stats' D_t needs the global sum)."""
        if self.group is None:
            return self._dual_wait(k)
        self._allreduce(self.pnq[1:])
        self._dual_phase(2, y, use_y, k)
        return self._dual_wait(k)

    def _dual_wait(self, k: int):
        tag = self._pending_tag[k]
        if self.stats_ev[k] is not None and tag is not None:
            tags = self.stats_host[k].numpy()[:, 5]
            spins, t0 = 0, None
            while not (tags == tag).all():
                spins += 1
                if spins & 0xFFF == 0:
                    import time
                    t0 = t0 or time.monotonic()
                    if time.monotonic() - t0 > 120.0:
                        raise _lib.RevsError("operator evaluation: timed out waiting for its stats")
            # (consumed: a later wait on this block without a new evaluation of ours -- the
            # block may since have been written by a native loop, which waits itself -- just reads)
            self._pending_tag[k] = None
        return self.stats_host[k].numpy().copy()

    def _dual_evaluate(self, y, use_y: bool, k: int, sup=None, kadd=None, dense=False):
        self._dual_launch(y, use_y, k, sup=sup, kadd=kadd, dense=dense)
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
        if o.native_newton and self._plan is not None and (self.group is None or self._comm is not None):
            return self._operator_solve_newton_native(first, pre)
        A = _lib.DUAL_AMAX
        scale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        self._fold_resume = False
        ycur, ytrial = self.yd
        cur = 0
        # (`_sup`: a candidate set that lists every row of the current multipliers, while there
        # are few enough of them for the row-wise home pass)
        stt = (self._dual_evaluate(ycur, self._y_support, cur, sup=self._sup)
               if first is None else first)
        evals, newton, pivots, ok_all = 1, 0, 0, False
        kadd_stt = o.newton_kadd         # what the evaluation behind `stt` admitted with (revs_plan_set_kadd_cold's rule)
        ns_prev, adm_prev = None, 0.0
        big = False                      # a slot needs more rows than a model of 128 takes: the lists of up to 512 go on
        best, stall = np.inf, 0
        last_small = False
        from_pre = pre is not None       # P_est_new is what `pre` (== `first` if nothing moved) wrote
        while True:
            if o.newton_trace:
                if not hasattr(self, "newton_trace"):
                    self.newton_trace = []
                self.newton_trace.append((self.iteration, newton, int(stt[:, 3].sum()), int(stt[:, 3].max()), int(stt[:, 2].sum()),
                                          int(stt[:, 2].max()), int(kadd_stt), float((stt[:, 0] / scale).max())))
            if (stt[:, 2] > A).any():
                big = True                               # more multipliers than a model of 128 rows holds
                break
            rmax = stt[:, 0] / scale
            if rmax.max() <= o.eps:
                ok_all = True
                break
            if newton >= o.newton_max:
                break
            # a slot whose model is full of multipliers while rows are still violated cannot
            # take them in; and a solve that stopped improving is not worth more iterations
            if ((stt[:, 2] >= A) & (stt[:, 3] > 0) & (rmax > o.eps)).any():
                big = True
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
            ncand = stt[:, 2] + np.minimum(stt[:, 3], np.minimum(kadd_stt, A - stt[:, 2]))
            self.model_calls[0 if ncand.max() <= 8 else 1] += 1
            last_small = bool(ncand.max() <= 8)
            few = stt[:, 2].max() + kadd_stt <= _lib.DUAL_FEW
            # rows to admit next (revs_plan_set_kadd_cold's rule): many, while a slot still shows many violated ones and either
            # the rows admitted last time nearly all kept a multiplier or some slot already carries 16 of them (rows that bind
            # one by one: the 121144 feeder); few, while a handful of multipliers clears hundreds of violated rows at once
            # (long laterals: the synthetic feeders, whose slots end with 3-4 multipliers)
            ns_sum = float(stt[:, 2].sum())
            adm_now = float(np.minimum(stt[:, 3], np.minimum(kadd_stt, A - stt[:, 2])).sum())
            kept = (ns_sum - ns_prev) / max(adm_prev, 1.0) if ns_prev is not None else 0.0
            kadd_next = (o.newton_kadd_cold if (o.newton_kadd_cold > o.newton_kadd and stt[:, 3].max() > o.newton_kadd_cold_at
                                                and (kept >= 0.5 or stt[:, 2].max() >= 16)) else o.newton_kadd)
            ns_prev, adm_prev = ns_sum, adm_now
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
                kadd_stn = o.newton_kadd if (use_pre and ls == 0) else kadd_next
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
                    stn = self._dual_evaluate(ytrial, True, nxt, sup=cur if few else None, kadd=kadd_next)
                evals += 1
                # (slack 1e-11 |D|: the evaluations sum the squares rounded to 2^-32 so that the sums do
                # not depend on their order -- a rounding of ~1e-13 |D| per evaluation)
                okk = stn[:, 1] >= D + 1e-4 * stn[:, 4] - 1e-11 * np.abs(D)
                pending &= ~okk
                if not pending.any():
                    break
                alpha[pending] *= 0.5
            pivots += int(np.abs(self.info_h.numpy()).sum())     # (the evaluation was waited for)
            if pending.any():
                break                                    # no ascent found: leave it to ADMM
            ycur, ytrial = ytrial, ycur
            cur, stt, kadd_stt = nxt, stn, kadd_stn
        self.yd = [ycur, ytrial]
        if big:
            return self._operator_solve_newton_big(newton, evals, pivots)
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
        self._sup = cur if (self._y_support and stt[:, 2].max() + o.newton_kadd <= _lib.DUAL_FEW) else None
        self.op_iters_hist.append(evals)
        self.op_path_hist.append("dual")
        self.op_converged = True
        return True

    def _ensure_big(self):
        """Buffers of the model problem beyond 128 rows per slot (csrc/newton_big.hip), on first use: lists, Gram slabs,
        Hessian and factor of up to REVS_DUAL_AMAX_BIG = 512 rows per slot in global memory (T = 24: ~300 MB)."""
        if getattr(self, "_big", None) is None:
            A2, T, dev = _lib.DUAL_AMAX_BIG, self.T, self.dev
            f64 = dict(dtype=torch.float64, device=dev)
            nks = int(min(4, max(1, self.M // 256)))
            self._big = dict(ci=torch.zeros(T, A2, dtype=torch.int64, device=dev), cc=torch.zeros(T, dtype=torch.int32, device=dev),
                             cv=torch.zeros(T, 3, A2, **f64), nks=nks, ks=torch.zeros(T, nks, A2, A2, **f64),
                             kf=torch.zeros(T, A2, A2, **f64), lf=torch.zeros(T, A2, A2, **f64), yh=torch.zeros(T, A2, **f64))
        return self._big

    def _operator_solve_newton_big(self, newton0=0, evals0=0, pivots0=0):
        """The Newton loop above on candidate lists of up to REVS_DUAL_AMAX_BIG = 512 rows per slot: where it goes on when a
        slot carries more multipliers than a model of 128 rows holds, or fills one with rows still violated (the
        reference hands Gurobi every row, lpsolver.py:183-194; the 121144 feeder needs 50-69, a feeder run closer to its
        limit more).  Same iteration -- lists, model maximised over the sign constraints by block principal pivoting,
        Armijo per slot, the same stopping test -- on kernels whose lists, Hessian and factor live in global memory
        (slow by construction: one workgroup per slot).  Evaluations go through the dense path (R p on the matrix cores:
        the row arrays the selection reads are then in memory).  False: more than 512 rows, no ascent, or no progress --
        the ADMM forms take the iteration, as before."""
        o, lib, M, T, st = self.op, self.lib, self.M, self.T, self.stream
        A2 = _lib.DUAL_AMAX_BIG
        scale = max(abs(self.vlo), abs(self.vhi), 1e-300)
        b = self._ensure_big()
        self._fold_resume = False
        self.big_solves = getattr(self, "big_solves", 0) + 1
        ycur, ytrial = self.yd
        stt = self._dual_evaluate(ycur, True, 0, dense=True)
        evals, newton, pivots, ok_all = evals0 + 1, newton0, pivots0, False
        best, stall, steps = np.inf, 0, 0
        while True:
            if (stt[:, 2] > A2).any():
                break
            rmax = stt[:, 0] / scale
            if rmax.max() <= o.eps:
                ok_all = True
                break
            if steps >= 4 * o.newton_max or ((stt[:, 2] >= A2) & (stt[:, 3] > 0) & (rmax > o.eps)).any():
                break
            if rmax.max() < 0.5 * best:
                best, stall = rmax.max(), 0
            else:
                stall += 1
                if stall >= 10:
                    break
            newton += 1
            steps += 1
            kadd = max(o.newton_kadd, o.newton_kadd_cold) if stt[:, 3].max() > o.newton_kadd_cold_at else o.newton_kadd
            check(lib.revs_op_dual_select_big(M, T, ptr(ycur), ptr(self.vfull), ptr(self.violw), self.vlo, self.vhi, int(kadd),
                                              ptr(b["ci"]), ptr(b["cc"]), ptr(b["cv"]), st), "revs_op_dual_select_big")
            check(lib.revs_op_dual_model_big(M, T, ptr(self.R64), ptr(self.pnq[1]), ptr(b["ci"]), ptr(b["cc"]), ptr(b["cv"]),
                                             self.kappa, o.newton_delta, o.newton_pivots, b["nks"], ptr(b["ks"]), ptr(b["kf"]),
                                             ptr(b["lf"]), ptr(b["yh"]), self.info_dev, st), "revs_op_dual_model_big")
            D = stt[:, 1]
            pending = rmax > o.eps
            alpha = pending.astype(np.float64)
            for ls in range(o.newton_ls):
                self.alpha_h.numpy()[:] = alpha
                check(lib.revs_op_dual_step_big(T, ptr(b["ci"]), ptr(b["cc"]), ptr(b["cv"]), ptr(b["yh"]), self.alpha_dev,
                                                ptr(ycur), M, ptr(ytrial), self.stats_dev[1] + 32, st), "revs_op_dual_step_big")
                stn = self._dual_evaluate(ytrial, True, 1, dense=True)
                evals += 1
                okk = stn[:, 1] >= D + 1e-4 * stn[:, 4] - 1e-11 * np.abs(D)
                pending &= ~okk
                if not pending.any():
                    break
                alpha[pending] *= 0.5
            pivots += int(np.abs(self.info_h.numpy()).sum())
            if pending.any():
                break
            # (the accepted trial's evaluation sits in stats block 1 and its row arrays in vfull / viol: the next lists are
            # built from them; the multipliers swap roles)
            ycur, ytrial = ytrial, ycur
            stt = stn
        self.yd = [ycur, ytrial]
        self.newton_hist.append((newton, evals, pivots))
        self._pre_kept = self._chain_ok = self._chain_few = self._spec_ok = False
        if not ok_all:
            ycur.zero_()
            self._y_support = False
            self._sup = None
            return False
        self._y_support = bool(stt[:, 2].sum() > 0)
        self._sup = None
        self.op_iters_hist.append(evals)
        self.op_path_hist.append("dual")
        self.op_converged = True
        return True

    def _operator_solve_newton_native(self, first, pre):
        """_operator_solve_newton's loop inside the library (revs_plan_newton_solve): same iterates and the
        same bookkeeping, one native call per operator solve.  `first` / `pre`: the caller's copies of stats blocks
        0 / 1; the library works from the pinned blocks themselves and is told the tags the caller saw, so that a
        launch that touched a block in between is an error, not a silently different solve."""
        have_first, have_pre = first is not None, pre is not None
        o = self.op
        self._fold_resume = False
        ys = (self.yd[0], self.yd[1])
        sup = self._sup if (self._y_support and self._sup is not None) else -1
        st = _lib.NewtonState(ptr(ys[0]), ptr(ys[1]), int(self._y_support), sup, ptr(self.P_est), ptr(self.P_sch),
                              ptr(self.G), ptr(self.P_est_new), int(have_first), int(have_pre), int(self._chain_few))
        # (one tag per evaluation, the same in every slot's record; a block whose slots carry different tags is not checked)
        st.first_tag = float(first[0, 5]) if have_first and (first[:, 5] == first[0, 5]).all() else 0.0
        st.pre_tag = float(pre[0, 5]) if have_pre and (pre[:, 5] == pre[0, 5]).all() else 0.0
        check(self.lib.revs_plan_newton_solve(self._plan, C.byref(st), self.stream), "revs_plan_newton_solve")
        self._pending_tag = [None, None]         # (the stats blocks were written by the native loop, which waited itself)
        self.yd = [ys[0], ys[1]] if st.y == ys[0].data_ptr() else [ys[1], ys[0]]
        self.model_calls[0] += st.models_small
        self.model_calls[1] += st.models_general
        if st.big_needed:                        # (y is where the loop stopped, not cleared)
            return self._operator_solve_newton_big(st.newton, st.evals, st.pivots)
        ok_all, newton = bool(st.ok), st.newton
        self.newton_hist.append((newton, st.evals, st.pivots))
        self._pre_kept = bool(st.pre_kept)
        self._chain_ok = bool(ok_all and newton == 1 and st.last_small and st.evals == 2)
        self._chain_few = bool(st.few)
        if ok_all and newton == 0:
            self._spec_wait = max(self._spec_wait - 1, 0)
            self._spec_ok = self._spec_wait == 0
        else:
            self._spec_ok = False
        if not ok_all:
            self._y_support = False
            self._sup = None
            return False
        self._y_support = st.nsup_sum > 0
        self._sup = st.cur if (self._y_support and st.nsup_max + o.newton_kadd <= _lib.DUAL_FEW) else None
        self.op_iters_hist.append(st.evals)
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
        tf = self._tree_newton and self._tree is not None          # rows by the tree form of R p, fused launches
        if self._tree_eval and self._tree is not None:             # (one block of partial sums per slot)
            nb = 1
        use_y = self._y_support
        if use_y and self._sup is not None:       # as _dual_launch, the selection left out
            self._dual_home_pass_rows(ycur, self._sup)
        else:
            self._dual_phase(1, ycur, use_y, 0)
        if self.group is not None:
            self._allreduce(self.pnq)
        if tf:       # rows, selection, small model and step of every slot in ONE launch
            check(lib.revs_op_dual_tree_select_model_step(
                M, T, C.byref(self._tree), ptr(self.pnq), ptr(ycur), self.vlo, self.vhi, o.newton_kadd,
                ptr(self.vfull), ptr(self.violw), ptr(self.d_part), ptr(self.c_idx[0]), ptr(self.c_cnt[0]),
                ptr(self.c_val[0]), self.stats_dev[0], 0.0, ptr(self.R64), self.kappa, o.newton_delta,
                o.newton_pivots, ptr(self.k_full), ptr(self.yhat), self.info_dev, scale, o.eps, ptr(ytrial),
                self.stats_dev[1] + 32, st), "revs_op_dual_tree_select_model_step")
        else:
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
        self._sup = 1 if (self._y_support and nmax + o.newton_kadd <= _lib.DUAL_FEW) else None
        self.op_iters_hist.append(2)
        self.op_path_hist.append("dual")
        self.op_converged = True

    def _fold_ok(self):
        """The folded chain applies (revs_plan_chain_fold_run): the native plan (sharded: with the library's
        own communicator in it -- the folded sums are exact and order-independent, one all-reduce of both
        arrays per iteration), the feeder as a tree the Newton evaluations use, few multipliers per slot
        (row-wise shifts), presolved PDHG."""
        # (the operator launch stages a slot's rows in LDS: 3 M doubles beside the tree's 16 KB and its own 30 KB)
        return (self._plan is not None and (self.group is None or self._comm is not None)
                and self._tree_newton and self._tree is not None
                and self.M <= _lib.CHAIN_FOLD_MAX_M and self._chain_few
                and self.op.chain_fold and not self.pdhg.full_rows)

    def _chain_run(self, count, write_sc=False):
        """Up to `count` iterations of the binding steady state inside one native call -- the folded
        chain (one pass over the residences and two launches per iteration) where it applies, else
        revs_plan_chain_run; the first iteration that is not the usual outcome is finished here as
        step() would.  Returns the number of iterations done (at least one)."""
        self._fused_ready = False
        self._p_clear = None
        if self._fold_ok():
            return self._chain_fold_run(count, write_sc)
        return self._chain_run_phases(count, write_sc)

    def _chain_run_phases(self, count, write_sc=False):
        """revs_plan_chain_run: the chained iteration as five launches and three passes over the
        residences (sharded runs, feeders without a tree, and the folded chain's second Newton step)."""
        self._fold_resume = False
        if self.group is not None:
            # sharded: ONE chained iteration issued in phases around the all-reduces of the node sums
            # (revs_plan_chain_run is the one-GPU loop), judged and booked as step() does
            self._chain_launch(write_sc, lambda i: None)
            if self._chain_accept():
                self.P_sch, self.P_sch_alt = self.P_sch_alt, self.P_sch
                self.G, self.G_alt = self.G_alt, self.G
                self.chain_hist[0] += 1
            else:
                self._chain_finish(False, 0, 0, write_sc)
            self.P_est, self.P_est_new = self.P_est_new, self.P_est
            self.iteration += 1
            return 1
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

    def _chain_fold_run(self, count, write_sc=False):
        if self._y_spare is None:
            self._y_spare = torch.zeros_like(self.yd[0])
        ys = (self.yd[0], self.yd[1], self._y_spare)
        # a third set of state buffers, borrowed from the pools the streaming loop rotates through: the next
        # iteration's sweep is enqueued unjudged behind the operator launch (roles rotate by the kept iterations)
        pool = self._state_pools()
        bufs = (self.P_est, self.P_est_new, self.P_sch, self.P_sch_alt, self.G, self.G_alt)
        third = (next(t for t in pool["pe"] if t is not self.P_est and t is not self.P_est_new),
                 next(t for t in pool["ps"] if t is not self.P_sch and t is not self.P_sch_alt),
                 next(t for t in pool["g"] if t is not self.G and t is not self.G_alt))
        sup0 = self._sup if (self._y_support and self._sup is not None) else -1
        st = _lib.ChainFoldState(ptr(ys[0]), ptr(ys[1]), ptr(ys[2]), int(self._y_support), sup0,
                                 *[ptr(t) for t in bufs], ptr(self.S) if write_sc else None,
                                 ptr(self.Csoc) if write_sc else None, int(self._fold_resume), 0, 0,
                                 *[ptr(t) for t in third])
        ypool = pool["y"]                # (the carried PDHG multipliers rotate with the state: a rejected sweep leaves them alone)
        if ypool is not None:
            ysp = [t for t in ypool if t is not self.pdhg_dual][:2]
            st.pdhg_dual, st.pdhg_dual_new, st.pdhg_dual_3 = ptr(self.pdhg_dual), ptr(ysp[0]), ptr(ysp[1])
        kept = C.c_int32()
        check(self.lib.revs_plan_chain_fold_run(self._plan, count, C.byref(st), C.addressof(kept), self.stream),
              "revs_plan_chain_fold_run")
        n = kept.value
        by = {t.data_ptr(): t for t in bufs + third}
        self.P_est, self.P_est_new = by[st.p_est], by[st.p_est_new]
        self.P_sch, self.P_sch_alt = by[st.p_sch], by[st.p_sch_alt]
        self.G, self.G_alt = by[st.gamma], by[st.gamma_alt]
        self.P_est_alt = by[st.p_est_3]      # (any pool member that is neither P_est nor P_est_new)
        if ypool is not None:
            self.pdhg_dual = next(t for t in ypool if t.data_ptr() == st.pdhg_dual)      # (the plan points at it already)
        yb = {t.data_ptr(): t for t in ys}
        self.yd = [yb[st.y], yb[st.y_trial]]
        self._y_spare = yb[st.y_spare]
        stepped = st.resume == 2          # stopped behind a good Newton step that needs another: y is that step
        self._fold_resume = st.resume == 1
        if n:
            self.model_calls[0] += n
            self.newton_hist.extend([(1, 2, -1)] * n)      # (pivot counts not read)
            self._pre_kept, self._chain_ok, self._spec_ok = True, True, False
            self._y_support = bool(st.use_y)
            self._sup = None         # (the lists of the accepted multipliers may sit in the plan's own sets)
            self.op_iters_hist.extend([2] * n)
            self.op_path_hist.extend(["dual"] * n)
            self.op_converged = True
            self.chain_hist[0] += n
            self.iteration += n
            if st.redone and st.resume == 1:   # the last kept iteration took more Newton steps inside the call
                self.fold_steps += st.redone
                self.newton_hist[-1] = (1 + st.redone, 2 + st.redone, -1)
                self.op_iters_hist[-1] = 2 + st.redone
        if n == count or (n and st.redone and not stepped and st.resume == 1):
            return n
        if st.redone or stepped:               # y is a Newton step the call has made: it carries multipliers
            self._y_support = True
        if stepped:
            # the operator's solve goes on from the step the chain has made: one chained iteration issued
            # in phases (evaluation, model, step, evaluation, sweep), the general loop behind it
            self._y_support, self._sup = True, None
            self.fold_steps += 1
            if not (write_sc and n == 0):      # (schedules are written by the general path only)
                done = self._chain_run_phases(1)
                self._book_step_before(st.pivots, 1 + st.redone)
                return n + done
        # the call stopped at an iteration that is the general loop's: a fresh solve from the current
        # multipliers (the folded chain's stats and lists are its own), then the sweep on its answer
        self._sup = None
        ok = self._operator_solve_newton()
        if ok and (stepped or st.redone):
            self._book_step_before(st.pivots if stepped else 0, (1 if stepped else 0) + st.redone)
        self.chain_hist[1] += 1
        if not ok:
            self._fast_cold = True
            self.op_cold = True
            self._require_converged(self.operator_solve(admm_only=True))
        self.agent_step(write_sc and n == 0)
        self.P_est, self.P_est_new = self.P_est_new, self.P_est
        self.iteration += 1
        return n + 1

    def _book_step_before(self, pivots, steps=1):
        """The solve just booked went on from `steps` Newton steps the folded chain had made: an
        iteration and an evaluation each (the step's own; the one at the old multipliers is the solve's
        first either way), and the last step's pivots, belong to it (earlier steps' are not read)."""
        if self.newton_hist:
            nw, ev, pv = self.newton_hist[-1]
            self.newton_hist[-1] = (nw + steps, ev + steps, pv + pivots if (pv >= 0 and steps == 1) else -1)
        if self.op_iters_hist:
            self.op_iters_hist[-1] += steps

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
