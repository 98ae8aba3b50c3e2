"""Kernels of the operator's dual Newton path, one by one through the C ABI, against their
numpy restatements in tests/fake_kernels.py (which follow include/revs_admm.h argument for
argument) and against the defining conditions of the model problem."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A = 128          # REVS_DUAL_AMAX


def _case(seed, n, M, T, n_mult):
    """A mid-solve state: float profiles, a radial-like PSD R, multipliers on n_mult rows of
    every slot (both signs, so that lower rows are exercised too)."""
    rng = np.random.default_rng(seed)
    node_of = np.sort(rng.integers(0, M, n))
    node_of[:M] = np.arange(M)                       # every node has a residence ...
    node_of = np.sort(node_of)
    if M > 3:
        node_of[node_of == 2] = 1                    # ... except node 2
    ptr = np.concatenate([[0], np.cumsum(np.bincount(node_of, minlength=M))]).astype(np.int64)
    B = rng.uniform(0, 1, (M, M)) * (rng.uniform(0, 1, (M, M)) < 0.2)
    R = (B @ B.T + np.diag(rng.uniform(0.5, 1.0, M))) * 1e-3
    pe = rng.uniform(0, 4, (n, T)).astype(np.float32)
    ps = rng.uniform(0, 4, (n, T)).astype(np.float32)
    gm = rng.normal(0, 3, (n, T)).astype(np.float32)
    y = np.zeros((M, T))
    for t in range(T):
        rows = rng.choice(M, min(n_mult, M), replace=False)
        y[rows, t] = rng.normal(0, 200.0, len(rows))
    return node_of, ptr, R, pe, ps, gm, y


def _run_evaluate(lib, fake, xp, host_ptr, M, T, ptr_, pe, ps, gm, R, y, kappa, vlo, vhi, kadd, ks):
    """revs_op_dual_evaluate on `lib` with arrays made by `xp` (torch-on-GPU or numpy);
    returns everything it writes, as numpy."""
    n = pe.shape[0]
    nblk = int(lib.revs_op_dual_blocks(M))
    mk = lambda shape, dt: xp(np.zeros(shape, dt))
    bufs = dict(d_sl=mk((ks, M, T), np.float64), v_sl=mk((ks, M, T), np.float64),
                pnq=mk((3, M, T), np.float64), pe_new=mk((n, T), np.float32),
                vfull=mk((M, T), np.float64), viol=mk((M, T), np.float64),
                part=mk((nblk, T, 4), np.float64), cidx=mk((T, A), np.int64),
                ccnt=mk((T,), np.int32), cval=mk((T, 3, A), np.float64),
                stats=mk((T, 8), np.float64))
    ins = dict(ptr=xp(ptr_), pe=xp(pe), ps=xp(ps), gm=xp(gm), R=xp(R),
               Rt=xp(np.ascontiguousarray(R.T)), y=xp(y))
    p = host_ptr
    rc = lib.revs_op_dual_evaluate(
        3, M, T, p(ins["ptr"]), p(ins["pe"]), p(ins["ps"]), p(ins["gm"]), p(ins["R"]), p(ins["Rt"]),
        p(ins["y"]), 1, kappa, vlo, vhi, kadd, ks, p(bufs["d_sl"]), p(bufs["v_sl"]), p(bufs["pnq"]),
        p(bufs["pe_new"]), p(bufs["vfull"]), p(bufs["viol"]), p(bufs["part"]), p(bufs["cidx"]),
        p(bufs["ccnt"]), p(bufs["cval"]), p(bufs["stats"]), 7.0, None, None)
    assert rc == 0
    return ins, bufs


@pytest.mark.parametrize("n,M,T,n_mult", [(3000, 300, 7, 5), (2500, 260, 24, 100), (900, 40, 96, 3)])
def test_dual_evaluate_and_model_match_numpy(gpu_lib, n, M, T, n_mult):
    import torch
    from fake_kernels import FakeKernels
    from revs_admm_amd._lib import check, ptr
    fake = FakeKernels()
    node_of, ptr_, R, pe, ps, gm, y = _case(n + T, n, M, T, n_mult)
    kappa, vlo, vhi, kadd, ks = 5.0, -0.05, 0.06, 16, 3
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    gi, gb = _run_evaluate(gpu_lib, fake, up, ptr, M, T, ptr_, pe, ps, gm, R, y, kappa, vlo, vhi, kadd, ks)
    keep = []                                         # numpy buffers must outlive the pointers
    def host(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return a
    ci, cb = _run_evaluate(fake, fake, host, lambda a: a.ctypes.data, M, T, ptr_, pe, ps, gm, R, y,
                           kappa, vlo, vhi, kadd, ks)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in gb.items()}
    # home pass: node sums, free counts (exact), dual value parts; P_est_new
    np.testing.assert_allclose(g["pnq"][0], cb["pnq"][0], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(g["pnq"][1], cb["pnq"][1])
    np.testing.assert_allclose(g["pnq"][2], cb["pnq"][2], rtol=1e-12, atol=1e-12)
    assert (g["pnq"][1][2] == 0).all() and (g["pnq"][0][2] == 0).all()      # the empty node
    np.testing.assert_allclose(g["pe_new"], cb["pe_new"], rtol=1e-6, atol=1e-7)
    assert (g["pe_new"] == 0).any() and (g["pe_new"] > 0).any()             # clamps are exercised
    # rows: v, residuals, D_t, counts
    np.testing.assert_allclose(g["vfull"], cb["vfull"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(g["stats"][:, :2], cb["stats"][:, :2], rtol=1e-10)
    np.testing.assert_array_equal(g["stats"][:, 2:4], cb["stats"][:, 2:4])
    assert (g["stats"][:, 5] == 7.0).all()                                 # the sequence tag
    assert (g["stats"][:, 2] == min(n_mult, M)).all() and (g["stats"][:, 3] > 0).any()
    # candidates: same rows in the same order, same signs, gradients, multipliers
    np.testing.assert_array_equal(g["ccnt"], cb["ccnt"])
    assert g["ccnt"].max() > (64 if n_mult > 64 else 0)                    # two-word sets exercised
    np.testing.assert_array_equal(g["cidx"], cb["cidx"])
    np.testing.assert_array_equal(g["cval"][:, 0], cb["cval"][:, 0])
    np.testing.assert_allclose(g["cval"][:, 1], cb["cval"][:, 1], rtol=1e-9, atol=1e-13)
    np.testing.assert_array_equal(g["cval"][:, 2], cb["cval"][:, 2])

    # ---- the model problem: Gram kernel + block principal pivoting ----
    nks, delta = 4, 1e-10
    kslab = torch.zeros(T, nks, A, A, dtype=torch.float64, device="cuda:0")
    kfull = torch.zeros(T, A, A, dtype=torch.float64, device="cuda:0")
    yhat = torch.zeros(T, A, dtype=torch.float64, device="cuda:0")
    info = torch.zeros(T, dtype=torch.int32, device="cuda:0")
    nfree = gb["pnq"][1].contiguous()
    check(gpu_lib.revs_op_dual_model(M, T, ptr(gi["R"]), ptr(nfree), ptr(gb["cidx"]), ptr(gb["ccnt"]),
                                     ptr(gb["cval"]), kappa, delta, 300, nks, ptr(kslab), ptr(kfull),
                                     ptr(yhat), ptr(info), None), "revs_op_dual_model")
    torch.cuda.synchronize()
    yh, inf, Kg = yhat.cpu().numpy(), info.cpu().numpy(), kfull.cpu().numpy()
    assert (inf >= 0).all()                                                # no pivot limit hit
    yh_ref, inf_ref = np.zeros((T, A)), np.zeros(T, np.int32)
    ksl_ref, kf_ref = np.zeros((T, nks, A, A)), np.zeros((T, A, A))
    Rc, Nc = np.ascontiguousarray(R), np.ascontiguousarray(cb["pnq"][1])
    fake.revs_op_dual_model(M, T, Rc.ctypes.data, Nc.ctypes.data, cb["cidx"].ctypes.data,
                            cb["ccnt"].ctypes.data, cb["cval"].ctypes.data, kappa, delta, 300, nks,
                            ksl_ref.ctypes.data, kf_ref.ctypes.data, yh_ref.ctypes.data,
                            inf_ref.ctypes.data, None)
    for t in range(T):
        a = int(g["ccnt"][t])
        if a <= 0:
            continue
        s, grad, ycur = g["cval"][t, 0, :a], g["cval"][t, 1, :a], g["cval"][t, 2, :a]
        RF = R[g["cidx"][t, :a]]
        K0 = (RF * cb["pnq"][1][:, t][None, :]) @ RF.T / kappa
        np.testing.assert_allclose(Kg[t, :a, :a], K0, rtol=1e-11, atol=1e-18)      # Gram kernel
        Kp = K0 * s[:, None] * s[None, :] + (delta * np.trace(K0) / a) * np.eye(a)
        c = s * grad + Kp @ np.maximum(s * ycur, 0.0)
        u = s * yh[t, :a]
        w = Kp @ u - c
        # the LCP that defines the model's maximiser over the sign constraints
        tol = 1e-9 * (np.abs(Kp).sum(axis=1).max() * u.max() + np.abs(c).max())
        assert u.min() >= 0.0
        assert w.min() >= -tol
        assert np.abs(w[u > 0]).max(initial=0.0) <= tol                 # complementarity
        np.testing.assert_allclose(yh[t, :a], yh_ref[t, :a], rtol=1e-4,
                                   atol=1e-5 * max(np.abs(yh_ref[t, :a]).max(), 1e-300))
        assert (yh[t, a:] == 0).all()

    # ---- step: y_trial at the candidates, exact; lin = grad . dy ----
    alpha = np.where(np.arange(T) % 3 == 0, 1.0, np.where(np.arange(T) % 3 == 1, 0.25, 0.0))
    d_alpha, ytrial = up(alpha), gi["y"].clone()
    lin = torch.zeros(T, 8, dtype=torch.float64, device="cuda:0")
    check(gpu_lib.revs_op_dual_step(T, ptr(gb["cidx"]), ptr(gb["ccnt"]), ptr(gb["cval"]), ptr(yhat),
                                    ptr(d_alpha), ptr(ytrial), ptr(lin), None), "revs_op_dual_step")
    torch.cuda.synchronize()
    yt, ln = ytrial.cpu().numpy(), lin.cpu().numpy()[:, 0]
    for t in range(T):
        a = max(int(g["ccnt"][t]), 0)
        rows = g["cidx"][t, :a]
        yo, yn = g["cval"][t, 2, :a], yh[t, :a]
        exp = yn if alpha[t] == 1.0 else (yo if alpha[t] == 0.0 else yo + alpha[t] * (yn - yo))
        np.testing.assert_array_equal(yt[rows, t], exp)
        other = np.setdiff1d(np.arange(M), rows)
        np.testing.assert_array_equal(yt[other, t], y[other, t])
        assert ln[t] == pytest.approx(float((g["cval"][t, 1, :a] * (exp - yo)).sum()), rel=1e-12, abs=1e-300)


def test_more_multipliers_than_the_model_holds(gpu_lib):
    """A slot with more than REVS_DUAL_AMAX multipliers is flagged (cand_cnt = -1), the others
    are served; the engine then hands the iteration to the ADMM forms."""
    import torch
    from fake_kernels import FakeKernels
    from revs_admm_amd._lib import ptr
    n, M, T = 1500, 200, 4
    node_of, ptr_, R, pe, ps, gm, y = _case(1, n, M, T, 10)
    y[:150, 1] = 1.0
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    gi, gb = _run_evaluate(gpu_lib, FakeKernels(), up, ptr, M, T, ptr_, pe, ps, gm, R, y, 5.0, -0.05,
                           0.06, 16, 2)
    torch.cuda.synchronize()
    cnt, st = gb["ccnt"].cpu().numpy(), gb["stats"].cpu().numpy()
    assert cnt[1] == -1 and st[1, 2] == (y[:, 1] != 0).sum() > A and (cnt[[0, 2, 3]] > 0).all()
