"""Operator side through the C ABI: matrix-core products, aggregation, and the
Utility QP (lpsolver.py:163-238) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _up(a, dt):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dt)).to("cuda:0")


@pytest.mark.parametrize("m,n,k", [(1126, 24, 1126), (37, 5, 19), (16, 16, 4), (1, 1, 1),
                                   (2048, 96, 2048), (130, 192, 257)])
def test_gemm_tn_f64(gpu_lib, m, n, k):
    """C = At^T B on v_mfma_f64_16x16x4_f64 vs numpy, incl. ragged edges.  Integer
    data first (exact: catches any fragment-layout slip), then random."""
    import torch
    from revs_admm_amd._lib import check, ptr
    rng = np.random.default_rng(m + n + k)
    for kind in ("int", "rand"):
        if kind == "int":
            At = rng.integers(-4, 5, (k, m)).astype(np.float64)
            B = rng.integers(-4, 5, (k, n)).astype(np.float64)
        else:
            At, B = rng.normal(size=(k, m)), rng.normal(size=(k, n))
        dA, dB = _up(At, np.float64), _up(B, np.float64)
        dC = torch.full((m, n), 7.0, dtype=torch.float64, device="cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        check(gpu_lib.revs_gemm_tn_f64(m, n, k, ptr(dA), m, ptr(dB), n, ptr(dC), n, 0, st))
        ref = At.T @ B
        got = dC.cpu().numpy()
        if kind == "int":
            assert (got == ref).all()
        else:
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * np.sqrt(k))
        check(gpu_lib.revs_gemm_tn_f64(m, n, k, ptr(dA), m, ptr(dB), n, ptr(dC), n, 1, st))
        np.testing.assert_allclose(dC.cpu().numpy(), 2 * ref, rtol=1e-12, atol=1e-11 * np.sqrt(k))


@pytest.mark.parametrize("m,T", [(1126, 24), (1126, 96), (100, 7)])
def test_voltage_f32(gpu_lib, m, T, feeder_R):
    """V = R P on v_mfma_f32_16x16x4_f32: the LinDistFlow sensitivity check
    (lpsolver.py:191-193).  Exact-f32 fma chains (4 K-quarters, summed) => error
    below 2e-6 * sum|r p| at K = 1126."""
    import torch
    from revs_admm_amd._lib import check, ptr
    rng = np.random.default_rng(T)
    R = feeder_R[:m, :m].astype(np.float32)
    P = rng.uniform(0, 8, (m, T)).astype(np.float32)
    # asymmetric integer check of the layout first
    Ai = rng.integers(-3, 4, (m, m)).astype(np.float32)
    Pi = rng.integers(-3, 4, (m, T)).astype(np.float32)
    dV = torch.zeros(m, T, dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    dAt, dPi, dR, dP = (_up(a, np.float32) for a in (Ai.T, Pi, R, P))   # keep alive
    check(gpu_lib.revs_voltage_f32(m, T, ptr(dAt), ptr(dPi), ptr(dV), st))
    assert (dV.cpu().numpy() == Ai @ Pi).all()
    check(gpu_lib.revs_voltage_f32(m, T, ptr(dR), ptr(dP), ptr(dV), st))
    ref = R.astype(np.float64) @ P.astype(np.float64)
    bound = 2e-6 * (np.abs(R).astype(np.float64) @ np.abs(P).astype(np.float64))
    assert (np.abs(dV.cpu().numpy() - ref) <= bound + 1e-12).all()


def test_aggregate(gpu_lib):
    import torch
    from revs_admm_amd._lib import check, ptr
    rng = np.random.default_rng(0)
    m, T = 37, 24
    counts = rng.integers(0, 6, m)
    counts[3] = 0
    n = counts.sum()
    node_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    x = rng.normal(size=(n, T))
    sc = rng.uniform(0.5, 2, m)
    out = torch.zeros(m, T, dtype=torch.float64, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    d_ptr, d_x, d_sc = _up(node_ptr, np.int64), _up(x, np.float64), _up(sc, np.float64)
    check(gpu_lib.revs_aggregate_f64(m, T, ptr(d_ptr), ptr(d_x), ptr(d_sc), ptr(out), st))
    ref = np.zeros((m, T))
    np.add.at(ref, np.repeat(np.arange(m), counts), x)
    np.testing.assert_allclose(out.cpu().numpy(), ref * sc[:, None], rtol=1e-14, atol=1e-14)


def _engine(w, mode="binary", **kw):
    from revs_admm_amd.engine import AdmmEngine
    return AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                      vlow=w.vlow, vhigh=w.vhigh, mode=mode, **kw)


def _set_state(e, pe, ps, gm):
    import torch
    for t, a in ((e.P_est, pe), (e.P_sch, ps), (e.G, gm)):
        t.copy_(torch.from_numpy(np.ascontiguousarray(a[e.perm], np.float32)))


@pytest.mark.parametrize("n,M,T", [(400, 50, 24), (300, 300, 24), (500, 40, 96)])
def test_operator_matches_oracle(gpu_lib, n, M, T):
    """Utility QP on the GPU (node-space SVD form, f64 matrix cores) == oracle
    (home-space dense form): the optimum is unique, so the two must agree."""
    from helpers import f32
    from oracle import revs_oracle as ro
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(n, T, n_nodes=M, seed=n + T, stress=1.3)
    rng = np.random.default_rng(1)
    ps = f32(w.load + rng.uniform(0, 3, (n, T)))
    if M == n:
        # the reference's case: every residence its own node.  Re-stress the feeder for
        # this assignment (30% over the limit at the worst node) so the rows still bind.
        w.node_of = np.arange(n)
        w.Rn = w.Rn * (1.3 * (w.vhigh ** 2 - w.vset ** 2) / (w.Rn @ ps).max())
    pe = f32(ps * rng.uniform(0.7, 1.1, (n, T)))
    gm = f32(rng.normal(0, 2.0, (n, T)))
    g0 = ro.utility_g0(pe, ps, gm, w.kappa)
    vlo, vhi = ro.voltage_limits(w.vset, w.vlow, w.vhigh)
    ref, info = ro.utility_solve(w.Rn, w.node_of, g0, w.kappa, vlo, vhi, return_info=True)
    assert np.abs(ref - np.maximum(g0, 0)).max() > 1e-2          # constraints bind
    # every operator path must give the oracle's answer: the dual Newton path (default),
    # the node-space ADMM fast path (taken when no residence is clamped at zero) and the
    # general home-space ADMM
    from revs_admm_amd.engine import OperatorOptions
    for solver, fast in (("admm", True), ("admm", False), ("newton-handoff", True), ("newton", True)):
        # "newton-handoff": a Newton solve that is not allowed to finish hands the iteration
        # to the ADMM forms, which must still deliver the answer
        e = _engine(w, op=(OperatorOptions(newton_max=1) if solver == "newton-handoff" else
                           OperatorOptions(solver=solver, node_fast=fast)))
        _set_state(e, pe, ps, gm)
        assert e.operator_solve()
        got = e.P_est_new.cpu().numpy()[e.inv_perm].astype(np.float64)
        assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), (fast, e.op_path_hist)
        if solver == "newton-handoff":
            assert e.newton_hist[-1][0] == 1 and e.op_path_hist[-1] in ("node", "home")
        elif solver == "newton":
            assert e.op_path_hist == ["dual"] and e.newton_hist[-1][0] >= 1
            # float64 all the way: the Newton answer is the oracle's to float32 rounding
            assert np.abs(got - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())
        elif not fast:
            assert e.op_path_hist == ["home"]
        elif ref.min() > 1e-6:
            assert e.op_path_hist == ["node"]
    # feasibility of the GPU answer itself, in double
    A = np.zeros((w.M, n)); A[w.node_of, np.arange(n)] = 1
    v = w.Rn @ (A @ got)
    assert v.max() <= vhi * (1 + 1e-5) and got.min() >= 0
    # second solve from the warm state converges immediately to the same answer
    it0 = e.op_iters_hist[-1]
    assert e.operator_solve() and e.op_iters_hist[-1] <= max(it0, 50)
    got2 = e.P_est_new.cpu().numpy()[e.inv_perm]
    assert np.abs(got2 - got).max() < 1e-5


def test_operator_golden_feeder(gpu_lib, golden, feeder_R):
    """Iteration 2 of the stored distributed run on the real 121144 feeder: the
    operator projects P_sch[1] (g0 = P_sch[1]) -- GPU vs oracle, and
    the KKT certificate of lpsolver.py:163-238 for the GPU answer."""
    from conftest import golden_homes
    from helpers import f32
    from oracle import revs_oracle as ro
    from revs_admm_amd.engine import AdmmEngine, pack_homes
    z, fd = golden
    oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
    n, T = oh.LOAD.shape
    zero = np.zeros((n, T))
    cost = f32(z["tariff_shift6"])
    oh.LOAD = f32(oh.LOAD)
    p, s, g1, st = ro.home_solve_binary(cost, oh, zero, zero, zero, 5.0)
    g1 = f32(g1)
    gm = f32(-2.5 * g1)                     # G[1] = -(kappa/2) P_sch[1]
    e = AdmmEngine(cost, pack_homes(oh.ev, 4.8, 20.0, 0.2, 11, 23), oh.LOAD, np.arange(n),
                   feeder_R, kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05)
    _set_state(e, zero, g1, gm)
    assert e.operator_solve()
    got = e.P_est_new.cpu().numpy().astype(np.float64)
    g0 = ro.utility_g0(zero, g1, gm, 5.0)
    vlo, vhi = ro.voltage_limits(1.03, 0.95, 1.05)
    ref, info = ro.utility_solve(feeder_R, np.arange(n), g0, 5.0, vlo, vhi, eps=1e-10,
                                 return_info=True)
    assert np.abs(got - ref).max() < 5e-5
    prim, stat, comp = ro.utility_kkt(feeder_R, np.arange(n), got, g0, 5.0, vlo, vhi,
                                      info["yv"], info["yb"])
    assert prim < 1e-6 * vhi + 1e-7 and stat < 1e-5
    # voltage check kernel on the answer: R.P <= vhi
    import torch
    v = e.voltage(e.P_est_new).cpu().numpy()
    assert v.max() <= vhi * (1 + 1e-4)


def test_gemm_x2_one_launch(gpu_lib):
    """Two products in one launch (how the operator issues V^T rhat | U^T w)."""
    import torch
    from revs_admm_amd._lib import check, ptr
    rng = np.random.default_rng(9)
    m, n = 333, 24          # odd m: the paired 16-byte A fetch falls back to scalar loads
    A0, A1 = rng.normal(size=(m, m)), rng.normal(size=(m, m))
    B0, B1 = rng.normal(size=(m, n)), rng.normal(size=(m, n))
    d = [_up(a, np.float64) for a in (A0, B0, A1, B1)]
    for ks in (1, 2, 5):
        C0 = torch.full((ks, m, n), 3.0, dtype=torch.float64, device="cuda:0")
        C1 = torch.full_like(C0, -1.0)
        check(gpu_lib.revs_gemm_tn_f64_x2(m, n, m, ptr(d[0]), ptr(d[1]), ptr(C0), ptr(d[2]),
                                          ptr(d[3]), ptr(C1), ks,
                                          torch.cuda.current_stream().cuda_stream))
        # the product is the sum of the K-split slabs
        np.testing.assert_allclose(C0.sum(0).cpu().numpy(), A0.T @ B0, rtol=1e-12, atol=1e-11)
        np.testing.assert_allclose(C1.sum(0).cpu().numpy(), A1.T @ B1, rtol=1e-12, atol=1e-11)


@pytest.mark.parametrize("m,T,ks", [(333, 24, 1), (2048, 24, 4), (1126, 24, 3), (257, 96, 2), (64, 7, 1)])
def test_gemm_cat(gpu_lib, m, T, ks):
    """[C0 | C1] = At^T [B0 | B1] with the split at column T (tiles straddle it)."""
    import torch
    from revs_admm_amd._lib import check, ptr
    rng = np.random.default_rng(m + T)
    A = rng.integers(-3, 4, (m, m)).astype(np.float64)
    B0 = rng.integers(-3, 4, (m, T)).astype(np.float64)
    B1 = rng.integers(-3, 4, (m, T)).astype(np.float64)
    dA, d0, d1 = _up(A, np.float64), _up(B0, np.float64), _up(B1, np.float64)
    C0 = torch.full((ks, m, T), 9.0, dtype=torch.float64, device="cuda:0")
    C1 = torch.full_like(C0, -9.0)
    check(gpu_lib.revs_gemm_tn_f64_cat(m, T, m, ptr(dA), ptr(d0), ptr(d1), ptr(C0), ptr(C1), ks,
                                       torch.cuda.current_stream().cuda_stream))
    assert (C0.sum(0).cpu().numpy() == A.T @ B0).all()
    assert (C1.sum(0).cpu().numpy() == A.T @ B1).all()


def test_operator_edge_topologies(gpu_lib):
    """One node carrying every residence; nodes without residences (their voltage rows
    drop out, as in the reference's R_res); a residence count of 1."""
    from helpers import f32
    from oracle import revs_oracle as ro
    from revs_admm_amd.synthetic import make_workload
    rng = np.random.default_rng(2)
    for n, M, node_of in ((40, 1, np.zeros(40, np.int64)),
                          (30, 12, np.sort(rng.choice([1, 4, 5, 9], 30))),
                          (1, 3, np.array([2]))):
        w = make_workload(n, 12, n_nodes=M, seed=n, stress=1.3)
        w.node_of = node_of.astype(np.int64)
        ps = f32(w.load + rng.uniform(0, 3, (n, 12)))
        pe = f32(ps * rng.uniform(0.7, 1.1, (n, 12)))
        gm = f32(rng.normal(0, 2.0, (n, 12)))
        A = np.zeros((M, n)); A[w.node_of, np.arange(n)] = 1
        used = np.unique(w.node_of)
        w.Rn = w.Rn * (1.3 * (w.vhigh ** 2 - w.vset ** 2) / (w.Rn[used] @ (A @ ps)).max())
        e = _engine(w)
        _set_state(e, pe, ps, gm)
        assert e.operator_solve()
        got = e.P_est_new.cpu().numpy()[e.inv_perm].astype(np.float64)
        g0 = ro.utility_g0(pe, ps, gm, w.kappa)
        vlo, vhi = ro.voltage_limits(w.vset, w.vlow, w.vhigh)
        # the engine constrains the nodes that carry residences
        sub = {m: i for i, m in enumerate(used)}
        ref = ro.utility_solve(w.Rn[np.ix_(used, used)], np.array([sub[m] for m in w.node_of]), g0,
                               w.kappa, vlo, vhi)
        assert np.abs(ref - np.maximum(g0, 0)).max() > 1e-3
        assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
