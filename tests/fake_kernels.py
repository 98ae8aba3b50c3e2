"""TEST DOUBLE for librevs_admm.so: the C ABI's entry points restated in numpy over
host memory, so that the driver in revs_admm_amd/engine.py (sharding, the all-reduce
of the node aggregate, stopping rules, rho adaptation) can be exercised under
`gloo` with world_size 2 on a machine without a GPU.  Lives in tests/, is never
imported by the product, and follows include/revs_admm.h / revs_admm_ops.h argument for argument.
The home solve delegates to the oracle."""
import ctypes as C

import numpy as np

from oracle import revs_oracle as ro
from revs_admm_amd._lib import DUAL_AMAX, HOME_DTYPE

_F = {np.float32: C.c_float, np.float64: C.c_double, np.int64: C.c_int64, np.int32: C.c_int32,
      np.uint8: C.c_uint8}


def view(p, shape, dt):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array((_F[dt] * n).from_address(int(p))).reshape(shape)


def g0f(pe, ps, gm, kappa):
    """revs_g0f of csrc/common.h: g0 = (P_est + P_sch)/2 - G/kappa in float -- the sum rounded
    once, then one fused multiply-add (emulated in extended precision, rounded to float once)."""
    inv = np.float32(1.0) / np.float32(kappa)
    s = np.float32(0.5) * (np.asarray(pe, np.float32) + np.asarray(ps, np.float32))
    r = s.astype(np.longdouble) - np.asarray(gm, np.float32).astype(np.longdouble) * np.longdouble(inv)
    return r.astype(np.float32).astype(np.float64)


class FakeKernels:
    def revs_pdhg_defaults(self, ref):
        o = ref._obj
        o.max_iter, o.check, o.tol, o.tau_scale, o.sigma_scale = 4000, 4, 1e-6, 0.0, 0.0

    def revs_residual_num_chunks(self, n):
        return min(256, (n + 4095) // 4096)

    def revs_agent_step(self, n, T, cost, homes, load, pe_old, pe_new, ps, gm, s_out, c_out, diff,
                        dsq, status, pdhg_dual, kappa, mode, pdhg, stream):
        return self.revs_agent_step_out(n, T, cost, homes, load, pe_old, pe_new, ps, gm, ps, gm,
                                        s_out, c_out, diff, dsq, status, pdhg_dual, kappa,
                                        mode, pdhg, stream)

    def revs_agent_step_out(self, n, T, cost, homes, load, pe_old, pe_new, ps, gm, ps_out, gm_out,
                            s_out, c_out, diff, dsq, status, pdhg_dual, kappa, mode, pdhg,
                            stream):
        f = lambda p, sh=(n, T): view(p, sh, np.float32)
        rec = view(homes, (n * HOME_DTYPE.itemsize,), np.uint8).view(HOME_DTYPE)
        oh = ro.homes_from_records(f(load).astype(float), rec)
        args = (f(cost, (T,)).astype(float), oh, f(pe_old).astype(float), f(ps).astype(float),
                f(gm).astype(float), kappa)
        p, s, g, st = (ro.home_solve_binary if mode == 0 else ro.home_solve_relaxed)(*args)
        chk = f(pe_new).astype(float) - g
        dg = g - f(ps).astype(float)
        f(gm_out)[:] = f(gm) + 0.5 * kappa * chk
        f(ps_out)[:] = g
        if s_out:
            f(s_out)[:] = p
        if c_out:
            f(c_out, (n, T + 1))[:] = s
        d = np.linalg.norm(chk, axis=1) / T
        view(diff, (n,), np.float32)[:] = d
        if status:
            view(status, (n,), np.int32)[:] = st
        view(dsq, (n,), np.float32)[:] = (dg ** 2).sum(axis=1)
        return 0

    def revs_agent_step_select(self, n, T, cost, homes, load, pe_old, pe_new, ps, gm, ps_out, gm_out,
                               s_out, c_out, diff, dsq, status, pdhg_dual, kappa, mode, pdhg, m,
                               sel_partial, y, vlo, vhi, kadd, vfull, viol, cidx, ccnt, cval, stats,
                               seq, node_of, p_next, pe_next, sel_nblk, stream):
        """The sweep; the selection itself was done by revs_op_dual_evaluate here (this double
        ignores phase bit 4), so only its sequence tag is left to write."""
        assert not p_next and not pe_next, "fused home pass: GPU only"
        rc = self.revs_agent_step_out(n, T, cost, homes, load, pe_old, pe_new, ps, gm, ps_out, gm_out,
                                      s_out, c_out, diff, dsq, status, pdhg_dual, kappa, mode, pdhg,
                                      stream)
        view(stats, (T, 8), np.float64)[:, 5] = seq
        return rc

    def revs_residual_finalize(self, diff, dsq, n, T, kappa, eps, scratch, out, stream):
        d = view(diff, (n,), np.float32).astype(float)
        q = view(dsq, (n,), np.float32).astype(float)
        o = view(out, (4,), np.float32)
        o[0] = np.sqrt(((d * T) ** 2).sum())
        o[1] = kappa * np.sqrt(q.sum())
        o[2] = d.max()
        o[3] = 1.0 if o[2] <= eps else 0.0
        return 0

    def revs_op_g0(self, n, T, pe, ps, gm, kappa, g0, stream):
        f = lambda p: view(p, (n, T), np.float32)
        view(g0, (n, T), np.float64)[:] = g0f(f(pe), f(ps), f(gm), kappa)
        return 0

    def revs_op_init_home(self, n, T, g0, sb, stream):
        view(sb, (n, T), np.float64)[:] = np.maximum(view(g0, (n, T), np.float64), 0)
        return 0

    def revs_op_init_node(self, m, T, cx, rho_v, bscale, vlo, vhi, zv, yv, w, stream):
        d = lambda p: view(p, (m, T), np.float64)
        bs = view(bscale, (m,), np.float64)[:, None] if bscale else 1.0
        d(zv)[:] = np.clip(d(cx), bs * vlo, bs * vhi)
        d(yv)[:] = 0
        d(w)[:] = view(rho_v, (T,), np.float64)[None, :] * d(zv)
        return 0

    @staticmethod
    def _seg(m, node_ptr):
        ptr = view(node_ptr, (m + 1,), np.int64)
        return np.repeat(np.arange(m), np.diff(ptr)), int(ptr[-1])

    def revs_aggregate_f64(self, m, T, node_ptr, inp, scale, out, stream):
        node, n = self._seg(m, node_ptr)
        acc = np.zeros((m, T))
        np.add.at(acc, node, view(inp, (n, T), np.float64))
        if scale:
            acc *= view(scale, (m,), np.float64)[:, None]
        view(out, (m, T), np.float64)[:] = acc
        return 0

    def revs_aggregate_f32(self, m, T, node_ptr, inp, out, stream):
        node, n = self._seg(m, node_ptr)
        acc = np.zeros((m, T), np.float32)
        np.add.at(acc, node, view(inp, (n, T), np.float32))
        view(out, (m, T), np.float32)[:] = acc
        return 0

    def revs_op_home_pass(self, m, T, node_ptr, isn, sb, g0, xc, rho_b, kappa, alpha, rhat,
                          cty, res, stream):
        node, n = self._seg(m, node_ptr)
        d = lambda p: view(p, (n, T), np.float64)
        rb = view(rho_b, (T,), np.float64)[None, :]
        isn_ = view(isn, (m,), np.float64)
        SB, G0 = d(sb), d(g0)
        Z, Y = np.maximum(SB, 0), np.minimum(SB, 0)
        if xc:
            xt = (kappa * G0 + rb * Z - Y) / (kappa + rb) + \
                (isn_[:, None] * view(xc, (m, T), np.float64))[node]
            u = alpha * xt + (1 - alpha) * Z + Y / rb
            Z, Y = np.maximum(u, 0), rb * np.minimum(u, 0)
            SB[:] = Z + Y
            if res:
                cy = (isn_[:, None] * view(cty, (m, T), np.float64))[node] + Y
                o = view(res, (8, T), np.float64)
                mx = lambda a: np.abs(a).max(axis=0) if a.shape[0] else np.zeros(T)
                for row, val in ((1, xt - Z), (2, kappa * (xt - G0) + cy), (5, xt), (6, cy),
                                 (7, kappa * G0)):
                    o[row] = np.maximum(o[row], mx(val))
        acc = np.zeros((m, T))
        np.add.at(acc, node, kappa * G0 + rb * Z - Y)
        view(rhat, (m, T), np.float64)[:] = isn_[:, None] * acc
        return 0

    def revs_op_home_pass_fused(self, m, T, node_ptr, isn, sb, g0, rho_b, kappa, alpha, rhat, nslab,
                                va, usa, rho_v, bscale, vlo, vhi, xc, zv, yv, w, stream):
        self.revs_op_node_update(m, T, nslab, va, rhat, usa, rho_v, rho_b, bscale, kappa, alpha,
                                 vlo, vhi, xc, zv, yv, w, None, stream)
        return self.revs_op_home_pass(m, T, node_ptr, isn, sb, g0, xc, rho_b, kappa, alpha, rhat,
                                      None, None, stream)

    def revs_op_row_scale(self, m, T, s, inp, out, stream):
        view(out, (m, T), np.float64)[:] = view(s, (m,), np.float64)[:, None] * \
            view(inp, (m, T), np.float64)
        return 0

    def revs_op_node_w(self, m, T, zv, yv, rho_v, w, stream):
        d = lambda p: view(p, (m, T), np.float64)
        d(w)[:] = view(rho_v, (T,), np.float64)[None, :] * d(zv) - d(yv)
        return 0

    def revs_op_node_scale(self, m, T, nslab, ta, tb, s, rho_v, rho_b, kappa, a, sa, stream):
        d = lambda p: view(p, (m, T), np.float64)
        ds = lambda p: view(p, (nslab, m, T), np.float64).sum(axis=0)
        sv = view(s, (m,), np.float64)[:, None]
        c = kappa + view(rho_b, (T,), np.float64)[None, :]
        av = (ds(ta) + sv * ds(tb)) / (c + view(rho_v, (T,), np.float64)[None, :] * sv * sv)
        d(a)[:] = av
        d(sa)[:] = sv * av
        return 0

    def revs_op_node_update(self, m, T, nslab, va, rhat, usa, rho_v, rho_b, bscale, kappa, alpha,
                            vlo, vhi, xc, zv, yv, w, res, stream):
        d = lambda p: view(p, (m, T), np.float64)
        rv = view(rho_v, (T,), np.float64)[None, :]
        c = kappa + view(rho_b, (T,), np.float64)[None, :]
        ds = lambda p: view(p, (nslab, m, T), np.float64).sum(axis=0)
        usa_ = ds(usa)
        d(xc)[:] = ds(va) - d(rhat) / c
        h = alpha * usa_ + (1 - alpha) * d(zv)
        bs = view(bscale, (m,), np.float64)[:, None] if bscale else 1.0
        zn = np.clip(h + d(yv) / rv, bs * vlo, bs * vhi)
        d(yv)[:] = d(yv) + rv * (h - zn)
        d(zv)[:] = zn
        d(w)[:] = rv * zn - d(yv)
        if res:
            o = view(res, (8, T), np.float64)
            for row, val in ((0, usa_ - zn), (3, usa_), (4, zn)):
                o[row] = np.maximum(o[row], np.abs(val).max(axis=0))
        return 0

    def revs_op_export(self, n, T, sb, pe, stream):
        view(pe, (n, T), np.float32)[:] = np.maximum(view(sb, (n, T), np.float64), 0)
        return 0

    def revs_gemm_tn_f64(self, m, n, k, At, lda, B, ldb, Cc, ldc, acc, stream):
        r = view(At, (k, lda), np.float64)[:, :m].T @ view(B, (k, ldb), np.float64)[:, :n]
        c = view(Cc, (m, ldc), np.float64)
        c[:, :n] = c[:, :n] + r if acc else r
        return 0

    def revs_gemm_tn_f64_x2(self, m, n, k, A0, B0, C0, A1, B1, C1, ksplit, stream):
        for A, B, Cc in ((A0, B0, C0), (A1, B1, C1)):
            a, b = view(A, (k, m), np.float64), view(B, (k, n), np.float64)
            c = view(Cc, (ksplit, m, n), np.float64)
            edges = np.linspace(0, k, ksplit + 1).astype(int)
            for q in range(ksplit):
                c[q] = a[edges[q]:edges[q + 1]].T @ b[edges[q]:edges[q + 1]]
        return 0

    def revs_gemm_tn_f64_cat(self, m, T, k, At, B0, B1, C0, C1, ksplit, stream):
        return self.revs_gemm_tn_f64_x2(m, T, k, At, B0, C0, At, B1, C1, ksplit, stream)

    def revs_gemm_tn_f64_split(self, m, n, k, At, B, Cc, ksplit, stream):
        a, b = view(At, (k, m), np.float64), view(B, (k, n), np.float64)
        c = view(Cc, (ksplit, m, n), np.float64)
        edges = np.linspace(0, k, ksplit + 1).astype(int)
        for q in range(ksplit):
            c[q] = a[edges[q]:edges[q + 1]].T @ b[edges[q]:edges[q + 1]]
        return 0

    # ---- node-space fast path ----
    def revs_op_node_prep(self, m, T, node_ptr, isn, pe, ps, gm, kappa, preclamp, p0, gmin, g0_out,
                          stream):
        node, n = self._seg(m, node_ptr)
        f = lambda p: view(p, (n, T), np.float32)
        g = g0f(f(pe), f(ps), f(gm), kappa)
        if g0_out:
            view(g0_out, (n, T), np.float64)[:] = g
        if preclamp:
            g = np.maximum(g, 0)
        acc = np.zeros((m, T))
        np.add.at(acc, node, g)
        mn = np.full((m, T), np.inf)
        np.minimum.at(mn, node, g)
        view(p0, (m, T), np.float64)[:] = view(isn, (m,), np.float64)[:, None] * acc
        view(gmin, (m, T), np.float64)[:] = mn
        return 0

    def revs_op_nodefast_feas(self, m, T, nslab, v0, bscale, gmin, vlo, vhi, cx, stats, stream):
        v = view(v0, (nslab, m, T), np.float64).sum(axis=0)
        bs = view(bscale, (m,), np.float64)[:, None] if bscale else 1.0
        view(cx, (m, T), np.float64)[:] = v
        st = view(stats, (2,), np.float64)
        st[0] = max(st[0], np.maximum(np.maximum(v - bs * vhi, bs * vlo - v), 0).max())
        st[1] = max(st[1], np.maximum(-view(gmin, (m, T), np.float64), 0).max())
        return 0

    def revs_op_nodefast_scale(self, m, T, nslab, wh, ph0, lam, rho_v, kappa, xh, sx, stream):
        d = lambda p: view(p, (m, T), np.float64)
        l = view(lam, (m,), np.float64)[:, None]
        w = view(wh, (nslab, m, T), np.float64).sum(axis=0)
        x = (kappa * d(ph0) + l * w) / (kappa + view(rho_v, (T,), np.float64)[None, :] * l * l)
        d(xh)[:] = x
        d(sx)[:] = l * x
        return 0

    def revs_op_nodefast_update(self, m, T, nslab, zt, rho_v, bscale, alpha, vlo, vhi, zv, yv, w,
                                res, stream):
        d = lambda p: view(p, (m, T), np.float64)
        ztv = view(zt, (nslab, m, T), np.float64).sum(axis=0)
        rv = view(rho_v, (T,), np.float64)[None, :]
        bs = view(bscale, (m,), np.float64)[:, None] if bscale else 1.0
        h = alpha * ztv + (1 - alpha) * d(zv)
        zn = np.clip(h + d(yv) / rv, bs * vlo, bs * vhi)
        d(yv)[:] = d(yv) + rv * (h - zn)
        d(zv)[:] = zn
        d(w)[:] = rv * zn - d(yv)
        if res:
            o = view(res, (8, T), np.float64)
            for row, val in ((0, ztv - zn), (3, ztv), (4, zn)):
                o[row] = np.maximum(o[row], np.abs(val).max(axis=0))
        return 0

    def revs_op_nodefast_dualres(self, m, T, nslab, xh, ph0, lam, yh, kappa, res, stream):
        d = lambda p: view(p, (m, T), np.float64)
        ly = view(lam, (m,), np.float64)[:, None] * view(yh, (nslab, m, T), np.float64).sum(axis=0)
        o = view(res, (8, T), np.float64)
        for row, val in ((2, kappa * (d(xh) - d(ph0)) + ly), (5, d(xh)), (6, ly), (7, kappa * d(ph0))):
            o[row] = np.maximum(o[row], np.abs(val).max(axis=0))
        return 0

    def revs_op_nodefast_finish(self, m, T, nslab, x, p0, gmin, isn, dd, slack, stats, stream):
        d = lambda p: view(p, (m, T), np.float64)
        dv = view(x, (nslab, m, T), np.float64).sum(axis=0) - d(p0)
        d(dd)[:] = dv
        d(slack)[:] = d(gmin) + view(isn, (m,), np.float64)[:, None] * dv
        st = view(stats, (2,), np.float64)
        st[0] = max(st[0], np.maximum(-d(slack), 0).max())
        st[1] = max(st[1], np.abs(d(p0)).max())
        return 0

    def revs_op_node_apply(self, m, T, node_ptr, isn, pe, ps, gm, kappa, preclamp, dd, pe_new,
                           stream):
        node, n = self._seg(m, node_ptr)
        f = lambda p: view(p, (n, T), np.float32)
        g = g0f(f(pe), f(ps), f(gm), kappa)
        if preclamp:
            g = np.maximum(g, 0)
        corr = (view(isn, (m,), np.float64)[:, None] * view(dd, (m, T), np.float64))[node]
        view(pe_new, (n, T), np.float32)[:] = np.maximum(g + corr, 0)
        return 0

    # ---- dual Newton path ----
    def revs_op_dual_eval(self, m, T, node_ptr, pe, ps, gm, nslab, dsl, kappa, pnq, pe_new, stream):
        node, n = self._seg(m, node_ptr)
        f = lambda p: view(p, (n, T), np.float32)
        g0 = g0f(f(pe), f(ps), f(gm), kappa)
        d = view(dsl, (nslab, m, T), np.float64).sum(axis=0) / kappa if dsl else np.zeros((m, T))
        dh = d[node]
        free = g0 > dh
        g = np.where(free, g0 - dh, 0.0)
        out = view(pnq, (3, m, T), np.float64)
        out[:] = 0.0
        np.add.at(out[0], node, g)
        np.add.at(out[1], node, free.astype(float))
        np.add.at(out[2], node, -0.5 * kappa * g * g)
        if pe_new:
            view(pe_new, (n, T), np.float32)[:] = g
        return 0

    def revs_op_dual_evaluate(self, phase, m, T, node_ptr, pe, ps, gm, R, Rt, y, use_y, kappa, vlo,
                              vhi, kadd, ksplit, d_sl, v_sl, pnq, pe_new, vfull, viol, partial,
                              cidx, ccnt, cval, stats, seq, tile_counters, stream):
        if phase & 1:
            if use_y:
                self.revs_gemm_tn_f64_split(m, T, m, R, y, d_sl, ksplit, stream)
            self.revs_op_dual_eval(m, T, node_ptr, pe, ps, gm, ksplit, d_sl if use_y else None,
                                   kappa, pnq, pe_new, stream)
        if phase & 2:
            self.revs_gemm_tn_f64_split(m, T, m, Rt, pnq, v_sl, ksplit, stream)
            self.revs_op_dual_select(m, T, ksplit, v_sl, pnq, y, vlo, vhi, kadd, vfull, viol,
                                     partial, cidx, ccnt, cval, stats, seq, stream)
        return 0

    def revs_op_dual_eval_rows(self, m, T, node_ptr, pe, ps, gm, R, sup_idx, sup_cnt, y, kappa, pnq,
                               pe_new, stream):
        Rm, yv = view(R, (m, m), np.float64), view(y, (m, T), np.float64)
        si, sc = view(sup_idx, (T, DUAL_AMAX), np.int64), view(sup_cnt, (T,), np.int32)
        d = np.zeros((1, m, T))
        for t in range(T):
            for i in range(int(sc[t])):
                f = int(si[t, i])
                d[0, :, t] += Rm[f, :] * yv[f, t]
        d = np.ascontiguousarray(d)
        return self.revs_op_dual_eval(m, T, node_ptr, pe, ps, gm, 1, d.ctypes.data, kappa, pnq, pe_new,
                                      stream)

    def revs_op_dual_blocks(self, m):
        return min(256, (m + 7) // 8)

    def revs_op_dual_select(self, m, T, nslab, vsl, pnq, y, vlo, vhi, kadd, vfull, viol, partial,
                            cidx, ccnt, cval, stats, seq, stream):
        A = DUAL_AMAX
        v = view(vsl, (nslab, m, T), np.float64).sum(axis=0)
        view(vfull, (m, T), np.float64)[:] = v
        yv = view(y, (m, T), np.float64)
        q = view(pnq, (3, m, T), np.float64)[2]
        ci, cc = view(cidx, (T, A), np.int64), view(ccnt, (T,), np.int32)
        cv, st = view(cval, (T, 3, A), np.float64), view(stats, (T, 8), np.float64)
        up = (yv > 0) | ((yv == 0) & (v > vhi))
        b = np.where(up, vhi, vlo)
        vi = np.maximum(np.maximum(v - vhi, vlo - v), 0.0)
        res = np.where(yv != 0, np.abs(v - b), vi)
        st[:, 0] = res.max(axis=0)
        st[:, 1] = (q - np.maximum(vhi * yv, vlo * yv)).sum(axis=0)
        st[:, 2] = (yv != 0).sum(axis=0)
        st[:, 3] = ((yv == 0) & (vi > 0)).sum(axis=0)
        st[:, 5] = seq
        ci[:] = 0
        cv[:] = 0.0
        cv[:, 0, :] = 1.0
        for t in range(T):
            sup = np.nonzero(yv[:, t])[0]
            if len(sup) > A:
                cc[t] = -1
                continue
            w = np.where(yv[:, t] == 0, vi[:, t], 0.0)
            order = sorted(np.nonzero(w > 0)[0], key=lambda r: (-w[r], r))
            rows = list(sup) + order[:min(kadd, A - len(sup))]
            cc[t] = len(rows)
            for i, r in enumerate(rows):
                ci[t, i] = r
                cv[t, 0, i] = 1.0 if up[r, t] else -1.0
                cv[t, 1, i] = v[r, t] - b[r, t]
                cv[t, 2, i] = yv[r, t]
        return 0

    def revs_op_dual_model(self, m, T, R, n_free, cidx, ccnt, cval, kappa, delta, max_pivots, nks,
                           k_slabs, k_full, yhat, info, stream):
        A = DUAL_AMAX
        Rm, Nf = view(R, (m, m), np.float64), view(n_free, (m, T), np.float64)
        ci, cc = view(cidx, (T, A), np.int64), view(ccnt, (T,), np.int32)
        cv, yh = view(cval, (T, 3, A), np.float64), view(yhat, (T, A), np.float64)
        inf = view(info, (T,), np.int32)
        for t in range(T):
            a = int(cc[t])
            if a <= 0:
                yh[t] = cv[t, 2]
                inf[t] = 0
                continue
            s = cv[t, 0, :a]
            RF = Rm[ci[t, :a]]
            K0 = (RF * Nf[:, t][None, :]) @ RF.T / kappa
            Kp = K0 * s[:, None] * s[None, :]
            Kp = Kp + (delta * np.trace(K0) / a + 1e-300) * np.eye(a)
            u = np.maximum(s * cv[t, 2, :a], 0.0)
            c = s * cv[t, 1, :a] + Kp @ u
            B = u > 0
            ninf, p, piv, done = a + 1, 3, 0, False
            while not done and piv < max_pivots:
                piv += 1
                u = np.zeros(a)
                idx = np.nonzero(B)[0]
                if len(idx):
                    u[idx] = np.linalg.solve(Kp[np.ix_(idx, idx)], c[idx])
                w = Kp @ u - c
                V = np.where(B, u < -1e-13 * np.abs(u).max(initial=0.0),
                             w < -1e-13 * np.abs(c).max(initial=0.0))
                nv = int(V.sum())
                if nv == 0:
                    done = True
                elif nv < ninf:
                    ninf, p = nv, 3
                    B = B ^ V
                elif p > 0:
                    p -= 1
                    B = B ^ V
                else:
                    i = np.nonzero(V)[0].max()
                    B[i] = not B[i]
            yh[t] = 0.0
            yh[t, :a] = s * np.maximum(u, 0.0)
            inf[t] = piv if done else -piv
        return 0

    def revs_op_dual_model_small(self, m, T, R, n_free, cidx, ccnt, cval, kappa, delta, max_pivots,
                                 k_full, yhat, info, stream):
        ks = np.zeros((T, 1, DUAL_AMAX, DUAL_AMAX))
        return self.revs_op_dual_model(m, T, R, n_free, cidx, ccnt, cval, kappa, delta, max_pivots, 1,
                                       ks.ctypes.data, k_full, yhat, info, stream)

    def revs_op_dual_step(self, T, cidx, ccnt, cval, yhat, alpha, ytrial, lin_out, stream):
        A = DUAL_AMAX
        ci, cc = view(cidx, (T, A), np.int64), view(ccnt, (T,), np.int32)
        cv, yh = view(cval, (T, 3, A), np.float64), view(yhat, (T, A), np.float64)
        al = view(alpha, (T,), np.float64)
        m = None
        for t in range(T):
            a, lin = max(int(cc[t]), 0), 0.0
            for i in range(a):
                yo = cv[t, 2, i]
                yn = yh[t, i] if al[t] == 1.0 else (yo if al[t] == 0.0 else yo + al[t] * (yh[t, i] - yo))
                C.c_double.from_address(int(ytrial) + 8 * (int(ci[t, i]) * T + t)).value = yn
                lin += cv[t, 1, i] * (yn - yo)
            C.c_double.from_address(int(lin_out) + 8 * 8 * t).value = lin
        return 0

    def revs_op_dual_step_pending(self, T, cidx, ccnt, cval, yhat, stats_prev, scale, eps, y, m,
                                  ytrial, lin_out, stream):
        if y:
            view(ytrial, (m, T), np.float64)[:] = view(y, (m, T), np.float64)
        st = view(stats_prev, (T, 8), np.float64)
        al = (st[:, 0] / scale > eps).astype(np.float64)
        return self.revs_op_dual_step(T, cidx, ccnt, cval, yhat, al.ctypes.data, ytrial, lin_out, stream)

    def revs_op_dual_select_model_step(self, m, T, sel_partial, sel_nblk, y, vlo, vhi, kadd, vfull, viol,
                                       cidx, ccnt, cval, stats, seq, R, n_free, kappa, delta,
                                       max_pivots, k_full, yhat, info, scale, eps, ytrial, lin_out,
                                       stream):
        """(the selection was done by revs_op_dual_evaluate here: this double ignores phase bit 4)"""
        view(stats, (T, 8), np.float64)[:, 5] = seq
        self.revs_op_dual_model_small(m, T, R, n_free, cidx, ccnt, cval, kappa, delta, max_pivots,
                                      k_full, yhat, info, stream)
        return self.revs_op_dual_step_pending(T, cidx, ccnt, cval, yhat, stats, scale, eps, y, m,
                                              ytrial, lin_out, stream)

    def revs_newton_chain_accept(self, T, s0, s1, scale, eps, amax, kadd, chain_few, nsup_sum, nsup_max):
        a, b = view(s0, (T, 8), np.float64), view(s1, (T, 8), np.float64)
        r0, r1 = a[:, 0] / scale, b[:, 0] / scale
        if (a[:, 2] > amax).any() or ((a[:, 2] >= amax) & (a[:, 3] > 0) & (r0 > eps)).any():
            return 0
        ncand = a[:, 2] + np.minimum(a[:, 3], np.minimum(kadd, amax - a[:, 2]))
        if not r0.max() > eps or ncand.max() > 8 or (a[:, 2].max() + kadd <= 48) != bool(chain_few):
            return 0
        pend = r0 > eps
        okk = b[:, 1] >= a[:, 1] + 1e-4 * b[:, 4] - 1e-11 * np.abs(a[:, 1])
        if (pend & ~okk).any() or (b[:, 2] > amax).any() or not r1.max() <= eps:
            return 0
        C.c_int32.from_address(int(nsup_sum)).value = int(b[:, 2].sum())
        C.c_int32.from_address(int(nsup_max)).value = int(b[:, 2].max())
        return 1

    def revs_voltage_f32(self, m, T, Rt, P, V, stream):
        view(V, (m, T), np.float32)[:] = view(Rt, (m, m), np.float32).T @ view(P, (m, T), np.float32)
        return 0

    def revs_last_error(self):
        return b""
