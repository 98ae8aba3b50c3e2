"""The operator QP (reference lpsolver.py:163-238) by ADMM in OSQP form -- the fallback of the
dual Newton path: home-space form, node-space fast path, rho calibration,
hipGraph replay of inner-iteration blocks.  Methods of AdmmEngine (mixed in by engine.py); every
number is computed by the kernels of csrc/operator_kernels.hip and csrc/gemm_kernels.hip."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr


class AdmmFormsMixin:
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
                with self._capture(g):
                    body()
                self._graph = g
            return self._graph.replay()
        if self._graph is None:
            gs = []
            for chk in (False, True):
                g = torch.cuda.CUDAGraph()
                with self._capture(g):
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
            with self._capture(g):
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
