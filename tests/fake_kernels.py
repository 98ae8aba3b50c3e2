"""TEST DOUBLE for librevs_admm.so: the C ABI's entry points restated in numpy over
host memory, so that the driver in revs_admm_amd/engine.py (sharding, the all-reduce
of the node aggregate, stopping rules, rho adaptation) can be exercised under
`gloo` with world_size 2 on a machine without a GPU.  Lives in tests/, is never
imported by the product, and follows include/revs_admm.h argument for argument.
The home solve delegates to the oracle."""
import ctypes as C

import numpy as np

from oracle import revs_oracle as ro
from revs_admm_amd._lib import HOME_DTYPE

_F = {np.float32: C.c_float, np.float64: C.c_double, np.int64: C.c_int64, np.int32: C.c_int32,
      np.uint8: C.c_uint8}


def view(p, shape, dt):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array((_F[dt] * n).from_address(int(p))).reshape(shape)


class FakeKernels:
    def revs_pdhg_defaults(self, ref):
        o = ref._obj
        o.max_iter, o.check, o.tol, o.tau_scale, o.sigma_scale = 4000, 8, 1e-6, 0.25, 4.0

    def revs_agent_num_partials(self, n, T):
        return (n + 31) // 32

    def revs_agent_step(self, n, T, cost, homes, load, pe_old, pe_new, ps, gm, s_out, c_out, diff,
                        partials, status, kappa, mode, pdhg, stream):
        f = lambda p, sh=(n, T): view(p, sh, np.float32)
        rec = view(homes, (n * HOME_DTYPE.itemsize,), np.uint8).view(HOME_DTYPE)
        oh = ro.homes_from_records(f(load).astype(float), rec)
        args = (f(cost, (T,)).astype(float), oh, f(pe_old).astype(float), f(ps).astype(float),
                f(gm).astype(float), kappa)
        p, s, g, st = (ro.home_solve_binary if mode == 0 else ro.home_solve_relaxed)(*args)
        chk = f(pe_new).astype(float) - g
        dg = g - f(ps).astype(float)
        f(gm)[:] = f(gm) + 0.5 * kappa * chk
        f(ps)[:] = g
        if s_out:
            f(s_out)[:] = p
        if c_out:
            f(c_out, (n, T + 1))[:] = s
        d = np.linalg.norm(chk, axis=1) / T
        view(diff, (n,), np.float32)[:] = d
        if status:
            view(status, (n,), np.int32)[:] = st
        npart = self.revs_agent_num_partials(n, T)
        part = view(partials, (npart, 3), np.float32)
        part[:] = 0
        part[0] = [(chk ** 2).sum(), (dg ** 2).sum(), d.max()]
        return 0

    def revs_residual_finalize(self, partials, npart, n, T, kappa, eps, out, stream):
        part = view(partials, (npart, 3), np.float32).astype(float)
        o = view(out, (4,), np.float32)
        o[0], o[1] = np.sqrt(part[:, 0].sum()), kappa * np.sqrt(part[:, 1].sum())
        o[2] = part[:, 2].max()
        o[3] = float(o[2] <= eps)
        return 0

    # ---- operator ----
    def revs_op_g0(self, n, T, pe, ps, gm, kappa, g0, stream):
        f = lambda p: view(p, (n, T), np.float32).astype(float)
        view(g0, (n, T), np.float64)[:] = 0.5 * (f(pe) + f(ps)) - f(gm) / kappa
        return 0

    def revs_op_init_home(self, n, T, g0, x, zb, yb, stream):
        d = lambda p: view(p, (n, T), np.float64)
        d(x)[:] = np.maximum(d(g0), 0)
        d(zb)[:] = d(x)
        d(yb)[:] = 0
        return 0

    def revs_op_init_node(self, m, T, cx, rho_v, vlo, vhi, zv, yv, w, stream):
        d = lambda p: view(p, (m, T), np.float64)
        d(zv)[:] = np.clip(d(cx), vlo, vhi)
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

    def revs_op_home_pass(self, m, T, node_ptr, isn, x, zb, yb, g0, xc, rho_b, kappa, sigma,
                          alpha, rhat, stream):
        node, n = self._seg(m, node_ptr)
        d = lambda p: view(p, (n, T), np.float64)
        rb = view(rho_b, (T,), np.float64)[None, :]
        isn_ = view(isn, (m,), np.float64)
        X, Z, Y, G0 = d(x), d(zb), d(yb), d(g0)
        if xc:
            c = kappa + sigma + rb
            rhs = sigma * X + kappa * G0 + rb * Z - Y
            xt = rhs / c + (isn_[:, None] * view(xc, (m, T), np.float64))[node]
            X[:] = alpha * xt + (1 - alpha) * X
            h = alpha * xt + (1 - alpha) * Z
            zn = np.maximum(h + Y / rb, 0)
            Y[:] = Y + rb * (h - zn)
            Z[:] = zn
        acc = np.zeros((m, T))
        np.add.at(acc, node, sigma * X + kappa * G0 + rb * Z - Y)
        view(rhat, (m, T), np.float64)[:] = isn_[:, None] * acc
        return 0

    def revs_op_node_w(self, m, T, zv, yv, rho_v, w, stream):
        d = lambda p: view(p, (m, T), np.float64)
        d(w)[:] = view(rho_v, (T,), np.float64)[None, :] * d(zv) - d(yv)
        return 0

    def revs_op_node_scale(self, m, T, ta, tb, s, rho_v, rho_b, kappa, sigma, a, sa, stream):
        d = lambda p: view(p, (m, T), np.float64)
        sv = view(s, (m,), np.float64)[:, None]
        c = kappa + sigma + view(rho_b, (T,), np.float64)[None, :]
        av = (d(ta) + sv * d(tb)) / (c + view(rho_v, (T,), np.float64)[None, :] * sv * sv)
        d(a)[:] = av
        d(sa)[:] = sv * av
        return 0

    def revs_op_node_update(self, m, T, va, rhat, usa, rho_v, rho_b, kappa, sigma, alpha, vlo,
                            vhi, xc, zv, yv, cx, w, stream):
        d = lambda p: view(p, (m, T), np.float64)
        rv = view(rho_v, (T,), np.float64)[None, :]
        c = kappa + sigma + view(rho_b, (T,), np.float64)[None, :]
        d(xc)[:] = d(va) - d(rhat) / c
        h = alpha * d(usa) + (1 - alpha) * d(zv)
        zn = np.clip(h + d(yv) / rv, vlo, vhi)
        d(yv)[:] = d(yv) + rv * (h - zn)
        d(zv)[:] = zn
        d(cx)[:] = alpha * d(usa) + (1 - alpha) * d(cx)
        d(w)[:] = rv * zn - d(yv)
        return 0

    def revs_op_residuals(self, m, T, node_ptr, isn, x, zb, yb, g0, cty, cx, zv, kappa, out,
                          stream):
        node, n = self._seg(m, node_ptr)
        d = lambda p: view(p, (n, T), np.float64)
        dm = lambda p: view(p, (m, T), np.float64)
        cy = (view(isn, (m,), np.float64)[:, None] * dm(cty))[node] + d(yb)
        o = view(out, (8, T), np.float64)
        mx = lambda a: np.abs(a).max(axis=0) if a.shape[0] else np.zeros(T)
        new = np.stack([mx(dm(cx) - dm(zv)), mx(d(x) - d(zb)), mx(kappa * (d(x) - d(g0)) + cy),
                        mx(dm(cx)), mx(dm(zv)), mx(d(x)), mx(cy), mx(kappa * d(g0))])
        o[:] = np.maximum(o, new)
        return 0

    def revs_op_export(self, n, T, zb, pe, stream):
        view(pe, (n, T), np.float32)[:] = view(zb, (n, T), np.float64)
        return 0

    def revs_gemm_tn_f64(self, m, n, k, At, lda, B, ldb, Cc, ldc, acc, stream):
        r = view(At, (k, lda), np.float64)[:, :m].T @ view(B, (k, ldb), np.float64)[:, :n]
        c = view(Cc, (m, ldc), np.float64)
        c[:, :n] = c[:, :n] + r if acc else r
        return 0

    def revs_gemm_tn_f64_x2(self, m, n, k, A0, B0, C0, A1, B1, C1, stream):
        self.revs_gemm_tn_f64(m, n, k, A0, m, B0, n, C0, n, 0, stream)
        self.revs_gemm_tn_f64(m, n, k, A1, m, B1, n, C1, n, 0, stream)
        return 0

    def revs_voltage_f32(self, m, T, Rt, P, V, stream):
        view(V, (m, T), np.float32)[:] = view(Rt, (m, m), np.float32).T @ view(P, (m, T), np.float32)
        return 0

    def revs_last_error(self):
        return b""
