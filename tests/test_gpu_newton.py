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
    # (round 3: p and the squares are summed rounded to 2^-36 / 2^-32, so that the sums do not depend
    # on their order: 1.5e-11 kW and 2.3e-10 kW^2 per residence and slot)
    np.testing.assert_allclose(g["pnq"][0], cb["pnq"][0], rtol=1e-12, atol=2e-9)
    np.testing.assert_array_equal(g["pnq"][1], cb["pnq"][1])
    np.testing.assert_allclose(g["pnq"][2], cb["pnq"][2], rtol=1e-12, atol=1e-7)
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

    # ---- the home pass with d from the listed rows of R instead of the dense product ----
    pnq_r = torch.zeros(3, M, T, dtype=torch.float64, device="cuda:0")
    pe_r = torch.zeros(n, T, dtype=torch.float32, device="cuda:0")
    check(gpu_lib.revs_op_dual_eval_rows(M, T, ptr(gi["ptr"]), ptr(gi["pe"]), ptr(gi["ps"]), ptr(gi["gm"]),
                                         ptr(gi["R"]), ptr(gb["cidx"]), ptr(gb["ccnt"]), ptr(gi["y"]), kappa,
                                         ptr(pnq_r), ptr(pe_r), None), "revs_op_dual_eval_rows")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(pnq_r[1].cpu().numpy(), g["pnq"][1])
    np.testing.assert_allclose(pnq_r[0].cpu().numpy(), g["pnq"][0], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(pnq_r[2].cpu().numpy(), g["pnq"][2], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(pe_r.cpu().numpy(), g["pe_new"], rtol=1e-6, atol=1e-6)

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
    """A slot with more than REVS_DUAL_AMAX multipliers is flagged (cand_cnt = -1) by the 128-row selection, the others
    are served; the engine's Newton loops then go on with lists of up to 512 rows (revs_op_dual_*_big:
    test_more_than_128_binding_rows_stay_on_the_newton_path)."""
    import torch
    from fake_kernels import FakeKernels
    from revs_admm_amd._lib import ptr
    n, M, T = 1500, 200, 4
    node_of, ptr_, R, pe, ps, gm, y = _case(1, n, M, T, 10)
    y[:150, 1] = 1.0
    y[:, 2] = 0.0
    y[40:168, 2] = -1.0                               # exactly REVS_DUAL_AMAX: served, no room left
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    gi, gb = _run_evaluate(gpu_lib, FakeKernels(), up, ptr, M, T, ptr_, pe, ps, gm, R, y, 5.0, -0.05,
                           0.06, 16, 2)
    torch.cuda.synchronize()
    cnt, st = gb["ccnt"].cpu().numpy(), gb["stats"].cpu().numpy()
    assert cnt[1] == -1 and st[1, 2] == (y[:, 1] != 0).sum() > A and (cnt[[0, 2, 3]] > 0).all()
    assert cnt[2] == A and (gb["cidx"].cpu().numpy()[2] == np.arange(40, 168)).all()
    assert (gb["cval"].cpu().numpy()[2, 0] == -1.0).all()


def test_product_rows_in_one_launch_equals_two_kernels(gpu_lib):
    """revs_op_dual_product_rows (rows epilogue in the last K-split workgroup of every row
    tile) against revs_gemm_tn_f64_split + revs_op_dual_rows: v and the violations bit for
    bit (same slab order), the per-slot folds of the partials equal, the cleared array
    cleared, the tile counters back at zero."""
    import torch
    from revs_admm_amd._lib import check, ptr
    M, T, ks = 300, 24, 4
    node_of, ptr_, R, pe, ps, gm, y = _case(3, 2500, M, T, 7)
    rng = np.random.default_rng(0)
    f64 = dict(dtype=torch.float64, device="cuda:0")
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    Rt, yd = up(R.T), up(y)
    pnq = up(np.stack([rng.uniform(0, 30, (M, T)), rng.integers(0, 9, (M, T)).astype(float),
                       -rng.uniform(0, 50, (M, T))]))
    vlo, vhi = -0.05, 0.06
    nblk_a, nblk_b = int(gpu_lib.revs_op_dual_blocks(M)), (M + 31) // 32
    out = {}
    for tag, nblk in (("two", nblk_a), ("one", nblk_b)):
        vs = torch.zeros(ks, M, T, **f64)
        vf, vi = torch.zeros(M, T, **f64), torch.zeros(M, T, **f64)
        part = torch.zeros(nblk, T, 4, **f64)
        zero = torch.full((M, T), 3.0, **f64)
        if tag == "two":
            check(gpu_lib.revs_gemm_tn_f64_split(M, T, M, ptr(Rt), ptr(pnq), ptr(vs), ks, None))
            check(gpu_lib.revs_op_dual_rows(M, T, ks, ptr(vs), ptr(pnq), ptr(yd), vlo, vhi, ptr(vf),
                                            ptr(vi), ptr(part), ptr(zero), None))
        else:
            cnt = torch.zeros(nblk, dtype=torch.int32, device="cuda:0")
            for _ in range(2):           # twice: the counters must reset themselves
                check(gpu_lib.revs_op_dual_product_rows(M, T, ptr(Rt), ptr(pnq), ptr(pnq), ptr(yd), vlo,
                                                        vhi, ks, ptr(vs), ptr(vf), ptr(vi), ptr(part),
                                                        ptr(zero), ptr(cnt), None))
            torch.cuda.synchronize()
            assert (cnt == 0).all()
        torch.cuda.synchronize()
        assert (zero == 0).all()
        out[tag] = (vf.cpu().numpy(), vi.cpu().numpy(), part.cpu().numpy())
    np.testing.assert_array_equal(out["one"][0], out["two"][0])
    np.testing.assert_array_equal(out["one"][1], out["two"][1])
    assert np.abs(out["one"][0] - R @ pnq[0].cpu().numpy()).max() < 1e-12 * np.abs(out["one"][0]).max()
    pa, pb = out["two"][2], out["one"][2]
    np.testing.assert_array_equal(pa[:, :, 0].max(0), pb[:, :, 0].max(0))
    np.testing.assert_allclose(pa[:, :, 1].sum(0), pb[:, :, 1].sum(0), rtol=1e-13)
    np.testing.assert_array_equal(pa[:, :, 2:].sum(0), pb[:, :, 2:].sum(0))
    assert pb[:, :, 2].sum() == (y != 0).sum() and pb[:, :, 3].sum() > 0


def test_product_rows_slab_handoff_under_load(gpu_lib):
    """The K-split slabs of revs_op_dual_product_rows are handed between workgroups inside the
    launch (sc1 stores, s_waitcnt vmcnt(0) in every storing wave, barrier, one agent-scope counter
    add; the last workgroup of a tile reads them with sc1 loads).  Stress it the way such
    hand-offs fail: 400 launches back to back at the benchmark shape (2048 nodes, 4 K-splits,
    every tile summed by a workgroup that did not store 3 of its 4 slabs), the slabs POISONED
    with NaN between launches so that a stale read cannot pass, a bandwidth-heavy kernel queued
    between them (uneven load), a different p every launch -- against the two-kernel path
    (product, then rows: a kernel boundary in between) on the same inputs, every word."""
    import torch
    from revs_admm_amd._lib import check, ptr
    M, T, ks, reps = 2048, 24, 4, 400
    rng = np.random.default_rng(5)
    f64 = dict(dtype=torch.float64, device="cuda:0")
    A = rng.uniform(0, 1, (M, 6))
    R = A @ A.T + np.eye(M)
    Rt = torch.from_numpy(R.T.copy()).to("cuda:0")
    y = torch.zeros(M, T, **f64)
    ps = [torch.from_numpy(np.stack([rng.uniform(0, 30, (M, T)), np.ones((M, T)), -np.ones((M, T))])).to("cuda:0")
          for _ in range(8)]
    vlo, vhi = -1e9, 1e9
    nblk = (M + 31) // 32
    vs = torch.zeros(ks, M, T, **f64)
    cnt = torch.zeros(nblk, dtype=torch.int32, device="cuda:0")
    vf = [torch.zeros(M, T, **f64) for _ in range(reps)]
    vi, part = torch.zeros(M, T, **f64), torch.zeros(nblk, T, 4, **f64)
    hog = torch.zeros(64 << 20, dtype=torch.float32, device="cuda:0")
    for r in range(reps):
        vs.fill_(float("nan"))                       # poison: a stale slab read shows up as NaN
        if r % 3 == 0:
            hog.add_(1.0)                            # uneven load on the memory system
        check(gpu_lib.revs_op_dual_product_rows(M, T, ptr(Rt), ptr(ps[r % 8][0]), ptr(ps[r % 8]), ptr(y), vlo,
                                                vhi, ks, ptr(vs), ptr(vf[r]), ptr(vi), ptr(part), None,
                                                ptr(cnt), None))
    torch.cuda.synchronize()
    assert (cnt == 0).all()
    ref = []
    vs2, vf2 = torch.zeros(ks, M, T, **f64), torch.zeros(M, T, **f64)
    part2 = torch.zeros(int(gpu_lib.revs_op_dual_blocks(M)), T, 4, **f64)
    for k in range(8):
        check(gpu_lib.revs_gemm_tn_f64_split(M, T, M, ptr(Rt), ptr(ps[k][0]), ptr(vs2), ks, None))
        check(gpu_lib.revs_op_dual_rows(M, T, ks, ptr(vs2), ptr(ps[k]), ptr(y), vlo, vhi, ptr(vf2),
                                        ptr(vi), ptr(part2), None, None))
        ref.append(vf2.clone())
    torch.cuda.synchronize()
    bad = [r for r in range(reps) if not torch.equal(vf[r], ref[r % 8])]
    assert not bad, f"stale or torn slab reads in launches {bad[:10]} (of {len(bad)})"


@pytest.mark.parametrize("mode", [1, 0])
def test_sweep_with_selection_and_next_home_pass(gpu_lib, mode):
    """revs_agent_step_select against its parts: the sweep's own outputs equal
    revs_agent_step_out's bit for bit; the selection riding in its launch equals
    revs_op_dual_select's; the folded home pass equals revs_op_dual_eval run afterwards on
    the new state (P_est candidate bit for bit, node sums to rounding: atomics)."""
    import torch
    from revs_admm_amd import _lib
    from revs_admm_amd._lib import check, ptr
    from revs_admm_amd.engine import pack_homes
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(5000, 24, n_nodes=120, seed=9, binary_feasible=(mode == 0), stress=1.0)
    n, T, M = w.N, 24, w.M
    rng = np.random.default_rng(4)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    f32 = dict(dtype=torch.float32, device="cuda:0")
    f64 = dict(dtype=torch.float64, device="cuda:0")
    order = np.argsort(w.node_of, kind="stable")
    homes = up(w.homes[order].view(np.uint8).reshape(n, _lib.HOME_DTYPE.itemsize))
    load, node32 = up(w.load[order].astype(np.float32)), up(w.node_of[order].astype(np.int32))
    cnt = np.bincount(w.node_of, minlength=M)
    nptr = up(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64))
    cost = up(w.cost.astype(np.float32))
    pe, pen = up(rng.uniform(0, 3, (n, T)).astype(np.float32)), up(rng.uniform(0, 3, (n, T)).astype(np.float32))
    ps, gm = up(rng.uniform(0, 3, (n, T)).astype(np.float32)), up(rng.normal(0, 1, (n, T)).astype(np.float32))
    pd = _lib.PDHG()
    gpu_lib.revs_pdhg_defaults(C.byref(pd))
    # selection inputs: a bookkeeping state with violations and a few multipliers
    y = np.zeros((M, T)); y[rng.choice(M, 5, replace=False), 3] = 50.0
    yd = up(y)
    vs = up(rng.uniform(0.0, 0.03, (2, M, T)))
    pnq = up(np.stack([rng.uniform(0, 30, (M, T)), np.ones((M, T)), -rng.uniform(0, 5, (M, T))]))
    nblk = int(gpu_lib.revs_op_dual_blocks(M))
    A_ = 128

    def sel_buffers():
        return dict(vf=torch.zeros(M, T, **f64), vi=torch.zeros(M, T, **f64),
                    part=torch.zeros(nblk, T, 4, **f64),
                    cidx=torch.zeros(T, A_, dtype=torch.int64, device="cuda:0"),
                    ccnt=torch.zeros(T, dtype=torch.int32, device="cuda:0"),
                    cval=torch.zeros(T, 3, A_, **f64), st=torch.zeros(T, 8, **f64))
    ref, fus = sel_buffers(), sel_buffers()
    check(gpu_lib.revs_op_dual_select(M, T, 2, ptr(vs), ptr(pnq), ptr(yd), -0.05, 0.06, 16, ptr(ref["vf"]),
                                      ptr(ref["vi"]), ptr(ref["part"]), ptr(ref["cidx"]), ptr(ref["ccnt"]),
                                      ptr(ref["cval"]), ptr(ref["st"]), 5.0, None))
    check(gpu_lib.revs_op_dual_rows(M, T, 2, ptr(vs), ptr(pnq), ptr(yd), -0.05, 0.06, ptr(fus["vf"]),
                                    ptr(fus["vi"]), ptr(fus["part"]), None, None))

    def outs():
        return dict(ps=torch.zeros(n, T, **f32), g=torch.zeros(n, T, **f32), diff=torch.zeros(n, **f32),
                    part=torch.zeros(n, **f32), status=torch.zeros(n, dtype=torch.int32, device="cuda:0"))
    a, b = outs(), outs()
    common = lambda o: (n, T, ptr(cost), ptr(homes), ptr(load), ptr(pe), ptr(pen), ptr(ps), ptr(gm),
                        ptr(o["ps"]), ptr(o["g"]), None, None, ptr(o["diff"]), ptr(o["part"]),
                        ptr(o["status"]), None, 5.0, mode, C.byref(pd))
    check(gpu_lib.revs_agent_step_out(*common(a), None))
    p_next, pe2 = torch.zeros(M, T, **f64), torch.zeros(n, T, **f32)
    check(gpu_lib.revs_agent_step_select(*common(b), M, ptr(fus["part"]), ptr(yd), -0.05, 0.06, 16,
                                         ptr(fus["vf"]), ptr(fus["vi"]), ptr(fus["cidx"]), ptr(fus["ccnt"]),
                                         ptr(fus["cval"]), ptr(fus["st"]), 5.0, ptr(node32), ptr(p_next),
                                         ptr(pe2), 0, None))
    torch.cuda.synchronize()
    for k in ("ps", "g", "diff", "part", "status"):
        assert torch.equal(a[k], b[k]), k
    for k in ("cidx", "ccnt", "cval"):
        assert torch.equal(ref[k], fus[k]), k
    assert torch.equal(ref["st"][:, :4], fus["st"][:, :4]) and (fus["st"][:, 5] == 5.0).all()
    assert (fus["ccnt"] > 0).any()
    # the folded home pass == revs_op_dual_eval on (P_est[k+1], P_sch[k+1], G[k+1])
    pnq2, pe_ref = torch.zeros(3, M, T, **f64), torch.zeros(n, T, **f32)
    check(gpu_lib.revs_op_dual_eval(M, T, ptr(nptr), ptr(pen), ptr(b["ps"]), ptr(b["g"]), 1, None, 5.0,
                                    ptr(pnq2), ptr(pe_ref), None))
    torch.cuda.synchronize()
    assert torch.equal(pe2, pe_ref) and (pe2 == 0).any() and (pe2 > 0).any()
    np.testing.assert_allclose(p_next.cpu().numpy(), pnq2[0].cpu().numpy(), rtol=1e-13, atol=2e-9)


@pytest.mark.parametrize("n_mult,kadd", [(3, 4), (1, 6), (6, 6)])
def test_selection_model_and_step_in_one_launch(gpu_lib, n_mult, kadd):
    """revs_op_dual_select_model_step against revs_op_dual_select + revs_op_dual_model_small +
    revs_op_dual_step_pending on the same rows: lists, stats, model answer, trial multipliers
    and linear terms bit for bit; slots with more than 8 candidates (n_mult + kadd > 8) get
    info = -999 and an unchanged column from both."""
    import torch
    from fake_kernels import FakeKernels
    from revs_admm_amd._lib import check, ptr
    n, M, T, ks = 2500, 260, 24, 3
    node_of, ptr_, R, pe, ps, gm, y = _case(11 + n_mult, n, M, T, n_mult)
    kappa, vlo, vhi = 5.0, -0.05, 0.06
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    gi, gb = _run_evaluate(gpu_lib, FakeKernels(), up, ptr, M, T, ptr_, pe, ps, gm, R, y, kappa, vlo, vhi,
                           kadd, ks)                    # (rows + the reference selection, seq 7)
    f64 = dict(dtype=torch.float64, device="cuda:0")
    nfree = gb["pnq"][1].contiguous()
    scale, eps = 0.06, 1e-8

    def outs():
        return dict(kfull=torch.zeros(T, A, A, **f64), yhat=torch.full((T, A), -3.0, **f64),
                    info=torch.full((T,), 77, dtype=torch.int32, device="cuda:0"),
                    ytr=torch.full((M, T), 9.0, **f64), lin=torch.zeros(T, 8, **f64))
    a, b = outs(), outs()
    check(gpu_lib.revs_op_dual_model_small(M, T, ptr(gi["R"]), ptr(nfree), ptr(gb["cidx"]), ptr(gb["ccnt"]),
                                           ptr(gb["cval"]), kappa, 1e-10, 300, ptr(a["kfull"]),
                                           ptr(a["yhat"]), ptr(a["info"]), None))
    check(gpu_lib.revs_op_dual_step_pending(T, ptr(gb["cidx"]), ptr(gb["ccnt"]), ptr(gb["cval"]),
                                            ptr(a["yhat"]), ptr(gb["stats"]), scale, eps, ptr(gi["y"]), M,
                                            ptr(a["ytr"]), ptr(a["lin"]), None))
    nblk = int(gpu_lib.revs_op_dual_blocks(M))
    cidx = torch.full((T, A), -5, dtype=torch.int64, device="cuda:0")
    ccnt = torch.full((T,), -5, dtype=torch.int32, device="cuda:0")
    cval, st = torch.full((T, 3, A), -5.0, **f64), torch.zeros(T, 8, **f64)
    check(gpu_lib.revs_op_dual_select_model_step(
        M, T, ptr(gb["part"]), nblk, ptr(gi["y"]), vlo, vhi, kadd, ptr(gb["vfull"]), ptr(gb["viol"]),
        ptr(cidx), ptr(ccnt), ptr(cval), ptr(st), 7.0, ptr(gi["R"]), ptr(nfree), kappa, 1e-10, 300,
        ptr(b["kfull"]), ptr(b["yhat"]), ptr(b["info"]), scale, eps, ptr(b["ytr"]), ptr(b["lin"]), None))
    torch.cuda.synchronize()
    assert torch.equal(cidx, gb["cidx"]) and torch.equal(ccnt, gb["ccnt"]) and torch.equal(cval, gb["cval"])
    assert torch.equal(st[:, [0, 1, 2, 3, 5]], gb["stats"][:, [0, 1, 2, 3, 5]])
    for k in ("yhat", "info", "ytr", "lin"):
        assert torch.equal(a[k], b[k]), k
    big = (ccnt > 8).cpu().numpy()
    assert big.any() == (n_mult + kadd > 8)
    assert (b["info"].cpu().numpy()[big] == -999).all() and (b["info"].cpu().numpy()[~big] != -999).all()
    yt, y0 = b["ytr"].cpu().numpy(), gi["y"].cpu().numpy()
    np.testing.assert_array_equal(yt[:, big], y0[:, big])
    assert big.all() or (yt[:, ~big] != y0[:, ~big]).any()


@pytest.mark.parametrize("a", [1, 2, 5, 8, 31, 63, 64, 65, 100, 127, 128])
def test_model_problem_sizes(gpu_lib, a):
    """revs_op_dual_model at every interesting candidate count (one row, the 64/65 boundary of
    the two mask words, the full 128): the returned point satisfies the LCP of the model and
    equals the numpy block-pivoting solution.  K = R R^T / kappa with R a x a, so K is the
    Gram matrix of the candidate rows themselves; two slots get mirrored signs."""
    import torch
    from fake_kernels import FakeKernels
    from revs_admm_amd._lib import check, ptr
    rng = np.random.default_rng(a)
    M, T, kappa, delta, nks = a, 3, 5.0, 1e-10, 2
    R = rng.normal(size=(M, M)) * (rng.uniform(size=(M, M)) < 0.5) + np.eye(M) * 0.1
    if a > 4:
        R[3] = R[1]                                    # an exactly repeated row: singular K
    nfree = np.ones((M, T))
    if a > 4:                                          # some nodes fully clamped in slot 1
        nfree[rng.integers(0, M, M // 5), 1] = 0.0
    cidx = np.tile(np.arange(A, dtype=np.int64), (T, 1)); cidx[:, a:] = 0
    ccnt = np.full(T, a, np.int32)
    cval = np.zeros((T, 3, A))
    sgn = np.where(rng.uniform(size=a) < 0.7, 1.0, -1.0)
    for t in range(T):
        s = sgn if t != 2 else -sgn
        cval[t, 0, :a] = s
        cval[t, 1, :a] = rng.normal(size=a)                       # gradient v - b
        cval[t, 2, :a] = s * np.maximum(rng.normal(size=a), 0.0)  # current y, right sign
    cval[:, 0, a:] = 1.0
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    dR, dN, dci, dcc, dcv = up(R), up(nfree), up(cidx), up(ccnt), up(cval)
    f64 = dict(dtype=torch.float64, device="cuda:0")
    ksl, kf = torch.zeros(T, nks, A, A, **f64), torch.zeros(T, A, A, **f64)
    yh, info = torch.zeros(T, A, **f64), torch.zeros(T, dtype=torch.int32, device="cuda:0")
    check(gpu_lib.revs_op_dual_model(M, T, ptr(dR), ptr(dN), ptr(dci), ptr(dcc), ptr(dcv), kappa, delta,
                                     400, nks, ptr(ksl), ptr(kf), ptr(yh), ptr(info), None), "model")
    torch.cuda.synchronize()
    yh, inf, Kg = yh.cpu().numpy(), info.cpu().numpy(), kf.cpu().numpy()
    assert (inf > 0).all()
    # a model without curvature (every residence behind its rows clamped) moves nothing
    y0, none_free = torch.zeros(T, A, **f64), torch.zeros_like(dN)
    check(gpu_lib.revs_op_dual_model(M, T, ptr(dR), ptr(none_free), ptr(dci), ptr(dcc), ptr(dcv),
                                     kappa, delta, 400, nks, ptr(ksl), ptr(kf), ptr(y0), ptr(info), None),
          "model")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(y0.cpu().numpy()[:, :a], cval[:, 2, :a])
    assert (info.cpu().numpy() == 0).all()
    # the one-kernel form for at most 8 candidates per slot: same Gram matrix, same solution
    kf2, yh2 = torch.zeros(T, A, A, **f64), torch.zeros(T, A, **f64)
    check(gpu_lib.revs_op_dual_model_small(M, T, ptr(dR), ptr(dN), ptr(dci), ptr(dcc), ptr(dcv), kappa,
                                           delta, 400, ptr(kf2), ptr(yh2), ptr(info), None), "small")
    torch.cuda.synchronize()
    if a <= 8:
        assert (info.cpu().numpy() > 0).all()
        np.testing.assert_allclose(kf2.cpu().numpy()[:, :a, :a], Kg[:, :a, :a], rtol=1e-11, atol=1e-15)
        np.testing.assert_allclose(yh2.cpu().numpy(), yh, rtol=1e-5, atol=1e-6 * np.abs(yh).max())
    else:                                   # too many candidates: flagged, nothing moved
        assert (info.cpu().numpy() == -999).all()
        np.testing.assert_array_equal(yh2.cpu().numpy()[:, :a], cval[:, 2, :a])
    ref, inf_ref = np.zeros((T, A)), np.zeros(T, np.int32)
    ksl_r, kf_r = np.zeros((T, nks, A, A)), np.zeros((T, A, A))
    Rc, Nc = np.ascontiguousarray(R), np.ascontiguousarray(nfree)
    FakeKernels().revs_op_dual_model(M, T, Rc.ctypes.data, Nc.ctypes.data, cidx.ctypes.data,
                                     ccnt.ctypes.data, cval.ctypes.data, kappa, delta, 400, nks,
                                     ksl_r.ctypes.data, kf_r.ctypes.data, ref.ctypes.data,
                                     inf_ref.ctypes.data, None)
    for t in range(T):
        s, grad, ycur = cval[t, 0, :a], cval[t, 1, :a], cval[t, 2, :a]
        K0 = (R * nfree[:, t][None, :]) @ R.T / kappa
        np.testing.assert_allclose(Kg[t, :a, :a], K0, rtol=1e-11, atol=1e-15)
        Kp = K0 * s[:, None] * s[None, :] + (delta * np.trace(K0) / a) * np.eye(a)
        c = s * grad + Kp @ np.maximum(s * ycur, 0.0)
        u = s * yh[t, :a]
        w = Kp @ u - c
        tol = 1e-8 * (np.abs(Kp).sum(axis=1).max() * max(u.max(), 1e-300) + np.abs(c).max())
        assert u.min() >= 0.0 and w.min() >= -tol and np.abs(w[u > 0]).max(initial=0.0) <= tol
        assert (yh[t, a:] == 0).all()
        obj = lambda z: 0.5 * z @ Kp @ z - c @ z                  # same optimal value as numpy's
        # (the repeated row with unrelated gradients leaves the delta shift in charge of a huge
        # component: objective values agree to its conditioning, not to rounding)
        assert obj(u) <= obj(s * ref[t, :a]) + 1e-5 * (abs(obj(s * ref[t, :a])) + 1e-300)


@pytest.mark.parametrize("case", ["synthetic2048", "golden1691", "tiny", "gaps", "synthetic4096", "synthetic8192",
                                  "synthetic3000", "synthetic16384", "synthetic12001"])
def test_tree_voltage_matches_dense_product(gpu_lib, case, golden, feeder_R):
    """revs_tree_voltage (R p as three prefix sums over the feeder in DFS preorder) == the dense
    float64 product Rn @ p at every checked row, to 1e-12 relative; rmax per slot == the largest
    violation; nodes without residences are neither injected nor checked."""
    import ctypes as C
    import torch
    from revs_admm_amd import _lib
    from revs_admm_amd.engine import feeder_tree, tree_voltage_host
    from revs_admm_amd.synthetic import make_workload
    rng = np.random.default_rng(7)
    if case == "golden1691":
        import networkx as nx
        from revs_admm_amd.lpsolver import feeder_arrays
        z, fd = golden
        g = nx.Graph()
        for nid, lab in zip(z["node_id"], fd.label):
            g.add_node(int(nid), label=lab.decode())
        for u, v, r in zip(fd.edge_u, fd.edge_v, fd.edge_r):
            g.add_edge(int(z["node_id"][u]), int(z["node_id"][v]), r=float(r))
        res = [n for n in g if g.nodes[n]["label"] == "H"]
        par, er, cons = feeder_arrays(g, res)
        Rn, T = feeder_R, 96
        checked = np.ones(len(res), bool)
    else:
        # (beyond 2048 nodes: workgroups of 512 x 8, 1024 x 8 and 1024 x 16 positions)
        M, T = {"synthetic2048": (2048, 24), "tiny": (1, 7), "gaps": (300, 33), "synthetic4096": (4096, 24),
                "synthetic8192": (8192, 5), "synthetic3000": (3000, 24), "synthetic16384": (16384, 3),
                "synthetic12001": (12001, 4)}[case]
        w = make_workload(max(M * 3, 10), 24, n_nodes=M, seed=4)
        par, er, cons = w.feeder
        Rn = w.Rn
        checked = np.ones(M, bool) if case != "gaps" else rng.random(M) < 0.6
    M = Rn.shape[0]
    tr = feeder_tree(par, er, cons, checked)
    p = rng.uniform(0.0, 4.0, (M, T)) * checked[:, None]
    ref = (Rn @ p) * checked[:, None]
    np.testing.assert_allclose(tree_voltage_host(tr, p), ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    dev = "cuda:0"
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = {"pack": up(tr["pack"].view(np.int64)), "w": up(tr["w"])}
    tree = _lib.Tree(tr["n"], d["pack"].data_ptr(), d["w"].data_ptr())
    dp, dv = up(p), torch.full((M, T), np.nan, dtype=torch.float64, device=dev)
    drm = torch.zeros(T, dtype=torch.float64, device=dev)
    vhi = float(np.quantile(ref[checked], 0.98))
    vlo = float(np.quantile(ref[checked], 0.01))
    _lib.check(gpu_lib.revs_tree_voltage(M, T, C.byref(tree), dp.data_ptr(), vlo, vhi, dv.data_ptr(),
                                         drm.data_ptr(), None), "revs_tree_voltage")
    torch.cuda.synchronize()
    v = dv.cpu().numpy()
    assert np.isnan(v[~checked]).all()                      # unchecked rows are not written
    np.testing.assert_allclose(v[checked], ref[checked], rtol=0, atol=1e-12 * np.abs(ref).max())
    want = np.maximum(np.maximum(ref - vhi, vlo - ref), 0.0)[checked].max(axis=0)
    np.testing.assert_allclose(drm.cpu().numpy(), want, rtol=0, atol=1e-12 * np.abs(ref).max())
    assert (want > 0).any()


@pytest.mark.parametrize("M,T", [(2048, 24), (4096, 24), (8192, 6), (12001, 5)])
def test_rows_by_the_tree_form_equal_the_dense_rows(gpu_lib, M, T):
    """The second half of a Newton evaluation -- voltages of the node sums, violations, the slot's four sums, the
    candidate lists -- by the tree form of R p (revs_op_dual_rows_tree: 256 x 8 positions per workgroup up to
    2048 nodes with the selection in the same launch; 512 x 8, 1024 x 8, 1024 x 16 beyond, round 4, with the
    selection behind it) against the dense f64 product on the matrix cores (revs_op_dual_evaluate, phase 2):
    same voltages to 1e-12 relative, same sums, same lists."""
    import ctypes as C
    import torch
    from revs_admm_amd import _lib
    from revs_admm_amd.engine import feeder_tree
    from revs_admm_amd.synthetic import make_workload
    rng = np.random.default_rng(M)
    w = make_workload(max(M * 3, 10), 24, n_nodes=M, seed=4)
    par, er, cons = w.feeder
    Rn = w.Rn
    tr = feeder_tree(par, er, cons, np.ones(M, bool))
    pnq = np.zeros((3, M, T))
    pnq[0] = rng.uniform(0.0, 4.0, (M, T))
    pnq[1] = rng.integers(0, 5, (M, T))
    pnq[2] = -rng.uniform(0.0, 30.0, (M, T))
    ref = Rn @ pnq[0]
    vhi, vlo = float(np.quantile(ref, 0.995)), float(np.quantile(ref, 0.002))
    y = np.zeros((M, T))
    for t in range(T):
        rows = rng.choice(M, 4, replace=False)
        y[rows, t] = rng.normal(0, 200.0, 4)
    kadd, ks, kappa = 6, 4, 5.0
    dev = "cuda:0"
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = {"pack": up(tr["pack"].view(np.int64)), "w": up(tr["w"])}
    tree = _lib.Tree(tr["n"], d["pack"].data_ptr(), d["w"].data_ptr())
    dpnq, dy, dRt = up(pnq), up(y), up(np.ascontiguousarray(Rn.T))
    nblk = int(gpu_lib.revs_op_dual_blocks(M))
    out = {}
    for which in ("dense", "tree"):
        z = lambda shape, dt=torch.float64: torch.zeros(shape, dtype=dt, device=dev)
        b = dict(vfull=z((M, T)), viol=z((M, T)), part=z((nblk, T, 4)), cidx=z((T, A), torch.int64),
                 ccnt=z((T,), torch.int32), cval=z((T, 3, A)), stats=z((T, 8)), v_sl=z((ks, M, T)))
        q = lambda t: t.data_ptr()
        if which == "dense":
            _lib.check(gpu_lib.revs_op_dual_evaluate(
                2, M, T, None, None, None, None, None, q(dRt), q(dy), 1, kappa, vlo, vhi, kadd, ks, None,
                q(b["v_sl"]), q(dpnq), None, q(b["vfull"]), q(b["viol"]), q(b["part"]), q(b["cidx"]), q(b["ccnt"]),
                q(b["cval"]), q(b["stats"]), 3.0, None, None), "revs_op_dual_evaluate")
        else:
            _lib.check(gpu_lib.revs_op_dual_rows_tree(
                M, T, C.byref(tree), q(dpnq), q(dy), vlo, vhi, kadd, q(b["vfull"]), q(b["viol"]), q(b["part"]), None,
                q(b["cidx"]), q(b["ccnt"]), q(b["cval"]), q(b["stats"]), 3.0, 1, None), "revs_op_dual_rows_tree")
        torch.cuda.synchronize()
        out[which] = {k: v.cpu().numpy() for k, v in b.items()}
    a_, b_ = out["dense"], out["tree"]
    scale = np.abs(ref).max()
    sa, sb = a_["stats"], b_["stats"]
    # the slot's sums: largest row residual, the dual value's terms, supports, violated rows; tag
    np.testing.assert_allclose(sb[:, 0], sa[:, 0], rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(sb[:, 1], sa[:, 1], rtol=1e-12)
    np.testing.assert_array_equal(sb[:, 2:4], sa[:, 2:4])
    assert (sb[:, 5] == 3.0).all() and (sa[:, 5] == 3.0).all()
    assert (sa[:, 3] > 0).all() and (sa[:, 2] == 4).all()
    np.testing.assert_array_equal(b_["ccnt"], a_["ccnt"])
    for t in range(T):
        n = int(a_["ccnt"][t])
        np.testing.assert_array_equal(b_["cidx"][t, :n], a_["cidx"][t, :n])
        np.testing.assert_allclose(b_["cval"][t, :, :n], a_["cval"][t, :, :n], rtol=0, atol=1e-12 * max(scale, 200.0))
    if M <= 2048:
        return                        # (rows staged in LDS: vfull / viol are scratch there)
    np.testing.assert_allclose(b_["vfull"], a_["vfull"], rtol=0, atol=1e-12 * scale)
    np.testing.assert_allclose(b_["viol"], a_["viol"], rtol=0, atol=1e-12 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("M,T", [(300, 24), (1126, 96), (2048, 24), (4096, 24), (8192, 6), (12001, 5)])
def test_shifts_by_the_tree_form_equal_the_dense_product(gpu_lib, M, T):
    """The first half of a Newton evaluation with multipliers: d = R^T y by the tree form of a radial feeder's R
    (revs_op_dual_evaluate_tree, phase 1: op_tree_shift_kernel in every shape of tree_body.h, round 5) against the dense
    f64 product on the matrix cores (revs_op_dual_evaluate): the same node sums p | N | q and the same answer P_est_new
    -- the shifts agree to 1e-12 of the largest; a residence whose g0 sits within that of its shift may land on the other
    side of the kink (none does on these inputs)."""
    import ctypes as C
    import torch
    from revs_admm_amd import _lib
    from revs_admm_amd.engine import feeder_tree
    from revs_admm_amd.synthetic import make_workload
    rng = np.random.default_rng(M + T)
    n = max(M * 3, 10)
    w = make_workload(n, 24, n_nodes=M, seed=4)
    par, er, cons = w.feeder
    Rn = w.Rn
    counts = np.bincount(w.node_of, minlength=M)
    tr = feeder_tree(par, er, cons, counts > 0)
    order = np.argsort(w.node_of, kind="stable")
    node_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    pe, ps, gm = (rng.uniform(0.0, 3.0, (n, T)).astype(np.float32)[order] for _ in range(3))
    y = np.zeros((M, T))
    have = np.flatnonzero(counts > 0)
    for t in range(T):
        rows = rng.choice(have, min(60, len(have)), replace=False)        # (more than REVS_DUAL_FEW rows: the product's case)
        y[rows, t] = rng.normal(0, 40.0, len(rows))
    kappa, ks = 5.0, 4
    dev = "cuda:0"
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = {"pack": up(tr["pack"].view(np.int64)), "w": up(tr["w"])}
    tree = _lib.Tree(tr["n"], d["pack"].data_ptr(), d["w"].data_ptr())
    dR, dy, dptr = up(Rn), up(y), up(node_ptr)
    dpe, dps, dgm = up(pe), up(ps), up(gm)
    out = {}
    for which in ("dense", "tree"):
        z = lambda shape, dt=torch.float64: torch.zeros(shape, dtype=dt, device=dev)
        b = dict(d_sl=z((ks, M, T)), pnq=z((3, M, T)), pen=z((n, T), torch.float32))
        q = lambda t: t.data_ptr()
        if which == "dense":
            _lib.check(gpu_lib.revs_op_dual_evaluate(
                1, M, T, q(dptr), q(dpe), q(dps), q(dgm), q(dR), None, q(dy), 1, kappa, -1.0, 1.0, 2, ks, q(b["d_sl"]),
                None, q(b["pnq"]), q(b["pen"]), None, None, None, None, None, None, None, 0.0, None, None), "revs_op_dual_evaluate")
        else:
            _lib.check(gpu_lib.revs_op_dual_evaluate_tree(
                1, M, T, q(dptr), q(dpe), q(dps), q(dgm), q(dR), C.byref(tree), q(dy), 1, kappa, -1.0, 1.0, 2, ks, q(b["d_sl"]),
                q(b["pnq"]), q(b["pen"]), None, None, None, None, None, None, None, 0.0, None), "revs_op_dual_evaluate_tree")
        torch.cuda.synchronize()
        out[which] = {k: v.cpu().numpy() for k, v in b.items()}
    ref = Rn.T @ y
    dd, dt = out["dense"]["d_sl"].sum(0), out["tree"]["d_sl"][0]
    occupied = counts > 0
    np.testing.assert_allclose(dd[occupied], ref[occupied], rtol=0, atol=1e-12 * np.abs(ref).max())
    np.testing.assert_allclose(dt[occupied], ref[occupied], rtol=0, atol=1e-12 * np.abs(ref).max())
    assert np.abs(ref[occupied]).max() > 0
    np.testing.assert_allclose(out["tree"]["pen"], out["dense"]["pen"], rtol=0, atol=5e-7)      # (float32 answers: one ulp)
    np.testing.assert_allclose(out["tree"]["pnq"][0], out["dense"]["pnq"][0], rtol=0, atol=1e-8)
    np.testing.assert_array_equal(out["tree"]["pnq"][1], out["dense"]["pnq"][1])
    assert (out["dense"]["pen"] > 0).any() and (out["dense"]["pen"] == 0).any()


@pytest.mark.gpu
def test_wavefront_reductions_keep_the_butterfly_bits(gpu_lib, tmp_path):
    """wave_sum_d (common.h: v_permlane32_swap / v_permlane16_swap + DPP row rotations) must give the xor
    butterfly's sums bit for bit -- the general loop's stand-alone kernels and the folded chain share it, and
    round 2's stored trajectories were made with the shuffles.  tools/probes/wave_reduce.hip carries both
    forms (the library's own header against the shuffles) and reports the mismatches over 65 536 random
    doubles of mixed magnitude, and wave_max_d / wave_min_i / wave_incl_scan_i against their shuffle forms."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tools", "probes", "wave_reduce.hip")
    exe = str(tmp_path / "wave_reduce")
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-I", os.path.join(root, "include"),
                    "-I", os.path.join(root, "revs_admm_amd", "csrc"), src, "-o", exe], check=True, capture_output=True)
    out = subprocess.run([exe], capture_output=True, text=True).stdout
    assert "mismatches 0 of 65536; max / min / scan disagreements 0;" in out, out


# ---- the model problem beyond REVS_DUAL_AMAX rows per slot (csrc/newton_big.hip) -----------------------------------------
A2 = 512         # REVS_DUAL_AMAX_BIG


@pytest.mark.parametrize("a", [5, 129, 200, 400, 512])
def test_model_problem_sizes_big(gpu_lib, a):
    """revs_op_dual_model_big at 129 / 200 / 400 / 512 candidate rows (and at 5: the path must not need many): the Gram matrix
    equals numpy's, the returned point satisfies the LCP of the model (u >= 0, K'u - c >= 0, complementary) and reaches the
    optimal value of the sign-constrained quadratic, which a non-negative least-squares solve on the Cholesky factor gives
    independently.  K = R N R^T / kappa with R a x a; an exactly repeated row makes K singular; slot 1 has clamped nodes,
    slot 2 mirrored signs."""
    import torch
    from scipy.optimize import nnls
    from revs_admm_amd._lib import check, ptr
    rng = np.random.default_rng(a)
    M, T, kappa, delta, nks = a, 3, 5.0, 1e-10, 3
    R = rng.normal(size=(M, M)) * (rng.uniform(size=(M, M)) < 0.5) + np.eye(M) * 0.1
    if a > 4:
        R[3] = R[1]
    nfree = np.ones((M, T))
    nfree[rng.integers(0, M, M // 5), 1] = 0.0
    cidx = np.tile(np.arange(A2, dtype=np.int64), (T, 1)); cidx[:, a:] = 0
    ccnt = np.full(T, a, np.int32)
    cval = np.zeros((T, 3, A2))
    sgn = np.where(rng.uniform(size=a) < 0.7, 1.0, -1.0)
    for t in range(T):
        s = sgn if t != 2 else -sgn
        cval[t, 0, :a] = s
        cval[t, 1, :a] = rng.normal(size=a)
        cval[t, 2, :a] = s * np.maximum(rng.normal(size=a), 0.0)
    cval[:, 0, a:] = 1.0
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    dR, dN, dci, dcc, dcv = up(R), up(nfree), up(cidx), up(ccnt), up(cval)
    f64 = dict(dtype=torch.float64, device="cuda:0")
    ksl, kf, lf = torch.zeros(T, nks, A2, A2, **f64), torch.zeros(T, A2, A2, **f64), torch.zeros(T, A2, A2, **f64)
    yh, info = torch.zeros(T, A2, **f64), torch.zeros(T, dtype=torch.int32, device="cuda:0")
    check(gpu_lib.revs_op_dual_model_big(M, T, ptr(dR), ptr(dN), ptr(dci), ptr(dcc), ptr(dcv), kappa, delta, 400, nks,
                                         ptr(ksl), ptr(kf), ptr(lf), ptr(yh), ptr(info), None), "model_big")
    torch.cuda.synchronize()
    yh, inf, Kg = yh.cpu().numpy(), info.cpu().numpy(), kf.cpu().numpy()
    assert (inf > 0).all(), inf
    for t in range(T):
        s, grad, ycur = cval[t, 0, :a], cval[t, 1, :a], cval[t, 2, :a]
        K0 = (R * nfree[:, t][None, :]) @ R.T / kappa
        np.testing.assert_allclose(Kg[t, :a, :a], K0, rtol=1e-11, atol=1e-13 * np.abs(K0).max())
        Kp = K0 * s[:, None] * s[None, :] + (delta * np.trace(K0) / a) * np.eye(a)
        c = s * grad + Kp @ np.maximum(s * ycur, 0.0)
        u = s * yh[t, :a]
        w = Kp @ u - c
        tol = 1e-8 * (np.abs(Kp).sum(axis=1).max() * max(u.max(), 1e-300) + np.abs(c).max())
        assert u.min() >= 0.0 and w.min() >= -tol and np.abs(w[u > 0]).max(initial=0.0) <= tol, (t, u.min(), w.min(), tol)
        assert (yh[t, a:] == 0).all()
        # an independent solve of min 1/2 u'K'u - c'u, u >= 0: with K' = L L^T it is the NNLS problem |L^T u - L^-1 c|
        Lc = np.linalg.cholesky(Kp + 1e-13 * np.trace(Kp) / a * np.eye(a))
        u_ref, _ = nnls(Lc.T, np.linalg.solve(Lc, c), maxiter=50 * a)
        obj = lambda z: 0.5 * z @ Kp @ z - c @ z
        assert obj(u) <= obj(u_ref) + 1e-6 * (abs(obj(u_ref)) + 1e-300), (t, obj(u), obj(u_ref))
    # a model without curvature moves nothing; a slot flagged -1 (more multipliers than 512) moves nothing and says so
    dcc2 = up(np.array([a, -1, a], np.int32))
    y0 = torch.zeros(T, A2, **f64)
    check(gpu_lib.revs_op_dual_model_big(M, T, ptr(dR), ptr(torch.zeros_like(dN)), ptr(dci), ptr(dcc2), ptr(dcv), kappa, delta,
                                         400, nks, ptr(ksl), ptr(kf), ptr(lf), ptr(y0), ptr(info), None), "model_big")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(y0.cpu().numpy()[[0, 2], :a], cval[[0, 2], 2, :a])
    assert info.cpu().numpy().tolist() == [0, -998, 0]


def test_select_and_step_big(gpu_lib):
    """revs_op_dual_select_big: the rows with a multiplier in row order, then the kadd most violated rows without one
    (larger violation first, ties to the lower row), signs and gradients from the row arrays -- 300 multipliers in one slot,
    more than 512 in another (flagged -1), none in a third; revs_op_dual_step_big: y_trial = y + alpha (yhat - y) on the
    listed rows only, gradient . step in lin_out."""
    import torch
    from revs_admm_amd._lib import check, ptr
    rng = np.random.default_rng(5)
    M, T, kadd, vlo, vhi = 2000, 4, 9, -0.05, 0.06
    y = np.zeros((M, T))
    y[rng.choice(M, 300, replace=False), 0] = rng.normal(0, 50, 300)
    y[rng.choice(M, 600, replace=False), 1] = 1.0
    y[rng.choice(M, 40, replace=False), 3] = -2.0
    v = rng.uniform(vlo, vhi, (M, T))
    viol = np.zeros((M, T))
    for t in (0, 2, 3):
        rows = rng.choice(np.flatnonzero(y[:, t] == 0), 25, replace=False)
        hi = rng.uniform(size=25) < 0.6
        amt = np.round(rng.uniform(1e-4, 1e-2, 25), 4)               # (repeated values: ties)
        v[rows, t] = np.where(hi, vhi + amt, vlo - amt)
        viol[rows, t] = amt
    up = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
    f64 = dict(dtype=torch.float64, device="cuda:0")
    dy, dv, dvi = up(y), up(v), up(viol)
    ci, cc, cv = torch.zeros(T, A2, dtype=torch.int64, device="cuda:0"), torch.zeros(T, dtype=torch.int32, device="cuda:0"), torch.zeros(T, 3, A2, **f64)
    check(gpu_lib.revs_op_dual_select_big(M, T, ptr(dy), ptr(dv), ptr(dvi), vlo, vhi, kadd, ptr(ci), ptr(cc), ptr(cv), None), "select_big")
    torch.cuda.synchronize()
    ci_h, cc_h, cv_h = ci.cpu().numpy(), cc.cpu().numpy(), cv.cpu().numpy()
    assert cc_h[1] == -1
    for t in (0, 2, 3):
        sup = np.flatnonzero(y[:, t] != 0)
        cand = np.flatnonzero((y[:, t] == 0) & (viol[:, t] > 0))
        order = cand[np.lexsort((cand, -viol[cand, t]))][:kadd]
        want = np.concatenate([sup, order])
        assert cc_h[t] == len(want) and (ci_h[t, :len(want)] == want).all(), t
        yv, vv = y[want, t], v[want, t]
        sg = np.where(yv != 0, np.sign(yv), np.where(vv > vhi, 1.0, -1.0))
        np.testing.assert_array_equal(cv_h[t, 0, :len(want)], sg)
        np.testing.assert_array_equal(cv_h[t, 1, :len(want)], vv - np.where(sg > 0, vhi, vlo))
        np.testing.assert_array_equal(cv_h[t, 2, :len(want)], yv)
        assert (cv_h[t, 0, len(want):] == 1).all() and (cv_h[t, 1:, len(want):] == 0).all()
    # the step
    yhat = up(rng.normal(0, 30, (T, A2)))
    alpha = up(np.array([1.0, 0.0, 0.5, 0.0]))
    ytr, lin = torch.full((M, T), 7.0, **f64), torch.zeros(T, 8, **f64)
    cc_fix = cc.clone(); cc_fix[1] = 0
    check(gpu_lib.revs_op_dual_step_big(T, ptr(ci), ptr(cc_fix), ptr(cv), ptr(yhat), ptr(alpha), ptr(dy), M, ptr(ytr), ptr(lin), None), "step_big")
    torch.cuda.synchronize()
    yt, yh_h, lin_h = ytr.cpu().numpy(), yhat.cpu().numpy(), lin.cpu().numpy()
    for t, al in ((0, 1.0), (2, 0.5), (3, 0.0)):
        n_ = cc_h[t]
        want = y[:, t].copy()
        rows = ci_h[t, :n_]
        want[rows] = yh_h[t, :n_] if al == 1.0 else (y[rows, t] if al == 0.0 else y[rows, t] + al * (yh_h[t, :n_] - y[rows, t]))
        np.testing.assert_array_equal(yt[:, t], want)
        np.testing.assert_allclose(lin_h[t, 0], (cv_h[t, 1, :n_] * (want[rows] - y[rows, t])).sum(), rtol=1e-12, atol=1e-300)
    np.testing.assert_array_equal(yt[:, 1], y[:, 1])


@pytest.mark.parametrize("vset,mode", [(1.043, "relaxed_exact"), (1.046, "binary")])
def test_more_than_128_binding_rows_stay_on_the_newton_path(gpu_lib, golden, feeder_R, vset, mode):
    """The reference's feeder run closer to its upper voltage limit (vset 1.043 / 1.046 instead of 1.03: 150 - 300 rows
    bind in a slot where 1.03 binds 50 - 69): lpsolver.py:183-194 hands Gurobi every row and gets an answer; here the
    operator's solves go on from the 128-row models to lists of up to 512 rows (csrc/newton_big.hip) instead of
    handing the iteration to the ADMM forms.  Three iterations against the oracle (its working-set dual solver, KKT-
    certified): diff, schedules, and the operator's last answer; the big path ran and the ADMM forms did not."""
    from conftest import golden_homes
    from helpers import f32
    from oracle import revs_oracle as ro
    from revs_admm_amd.engine import AdmmEngine, pack_homes
    z, fd = golden
    oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
    n, T = oh.LOAD.shape
    cost = f32(z["tariff_shift6"])
    oh = ro.Homes.uniform(f32(oh.LOAD), oh.ev, 4.8, 20.0, 0.2, 11, 23)
    e = AdmmEngine(cost, pack_homes(oh.ev, 4.8, 20.0, 0.2, 11, 23), oh.LOAD, np.arange(n), feeder_R, kappa=5.0, vset=vset,
                   vlow=0.95, vhigh=1.05, mode=mode)
    iters = 3
    d = e.run(iters)
    P, S, Cs = e.result()
    assert set(e.op_path_hist) == {"dual"} and getattr(e, "big_solves", 0) >= 1
    nsup = int((e.yd[0].cpu().numpy() != 0).sum(0).max())
    omode = "relaxed" if mode == "relaxed_exact" else "binary"
    d_ref, P_ref, S_ref, C_ref, tr = ro.solve_ADMM(oh, feeder_R, np.arange(n), cost, 5.0, iters, vset, 0.95, 1.05, mode=omode,
                                                   util_method="dual", keep=True)
    pe = e.P_est.cpu().numpy()[e.inv_perm]
    print(f"vset {vset}, {mode}: rows with a multiplier in the fullest slot {nsup}, big solves {e.big_solves}, newton "
          f"{[h[0] for h in e.newton_hist]}, evaluations {e.op_iters_hist}; |P_est - oracle| {np.abs(pe - tr.P_est[-1]).max():.2e} kW, "
          f"|diff - oracle| {np.abs(d - d_ref).max():.2e}")
    assert nsup > 128
    assert np.abs(d[:2] - d_ref[:2]).max() < 1e-3 * max(1.0, d_ref.max())
    if mode == "relaxed_exact":
        assert np.abs(d - d_ref).max() < 1e-3 * max(1.0, d_ref.max())
        assert np.abs(pe - tr.P_est[-1]).max() < 1e-4 and np.abs(S - S_ref).max() < 1e-4
