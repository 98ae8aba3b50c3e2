"""Whole ADMM loop (lpsolver.py:242-290) on the GPU vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(w, mode, **kw):
    """The synthetic feeder is handed over as a tree too (OperatorOptions.voltage = "auto"): the
    steady state then runs as one launch per iteration with the rows judged by the tree form of
    R p; voltage="dense" keeps the matrix-core product."""
    from revs_admm_amd.engine import AdmmEngine
    kw.setdefault("feeder", w.feeder)
    return AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                      vlow=w.vlow, vhigh=w.vhigh, mode=mode, **kw)


@pytest.mark.parametrize("solver", ["newton", "admm"])
@pytest.mark.parametrize("stress", [0.9, 1.5])
@pytest.mark.parametrize("mode,omode", [("relaxed_exact", "relaxed"), ("pdhg", "relaxed"), ("pdhg_presolve", "relaxed")])
def test_relaxed_trajectory(gpu_lib, mode, omode, stress, solver):
    """Continuous homes: the iteration map is Lipschitz, so the whole trajectory must
    follow the oracle.  Tolerance: 2e-5 kW on schedules (measured 2e-6 ... 4e-6, both operator solvers; was 2e-3),
    1e-3 relative on diff (measured 6e-7 absolute).
    (pdhg_presolve: revs_pdhg_t::polish = 3, the KKT steps from the carried multiplier before PDHG.)"""
    from helpers import f32, oracle_homes
    from oracle import revs_oracle as ro
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(600, 24, n_nodes=60, seed=11, binary_feasible=False, stress=stress)
    w.load, w.cost = f32(w.load), f32(w.cost)
    iters = 8
    from revs_admm_amd.engine import OperatorOptions
    kw = {}
    if mode == "pdhg_presolve":
        mode, kw = "pdhg", dict(pdhg=dict(polish=3))
    e = _engine(w, mode, op=OperatorOptions(solver=solver), **kw)
    diffs = e.run(iters)
    P_sch, S, C = e.result()
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oracle_homes(w), w.Rn, w.node_of, w.cost, w.kappa,
                                               iters, w.vset, w.vlow, w.vhigh, mode=omode,
                                               util_eps=1e-10)
    print(f"trajectory {mode}/{solver}: max |diff - oracle| {np.abs(diffs - d_ref).max():.2e}, |S - oracle| "
          f"{np.abs(S - S_ref).max():.2e} kW, |C - oracle| {np.abs(C - C_ref).max():.2e}")
    assert np.abs(diffs - d_ref).max() < 1e-3 * max(1.0, d_ref.max())
    tol = 2e-5                                      # (5 x the measured error; was 2e-3)
    assert np.abs(S - S_ref).max() < tol
    assert np.abs(P_sch - P_ref).max() < tol
    assert np.abs(C - C_ref).max() < 5e-6
    # the run is doing real work: operator rows bind, and the first iterations (where some
    # g0 go negative) pass through the general home-space path before the fast path resumes
    assert d_ref[-1].mean() < d_ref[0].mean()
    if solver == "newton":
        assert set(e.op_path_hist) == {"dual"} and max(n for n, _, _ in e.newton_hist) >= 1
        assert e.P_est.min().item() == 0.0 or stress < 1.0
    else:
        assert max(e.op_iters_hist) >= 25 and {"node", "home"} <= set(e.op_path_hist)


def test_binary_teacher_forced(gpu_lib):
    """Binary homes: one flipped near-tie changes every later iterate, so each
    iteration is checked from the ORACLE's state (teacher forcing): operator output,
    schedules where the oracle's choice is clear, dual update."""
    import torch
    from helpers import f32, oracle_homes
    from oracle import revs_oracle as ro
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(500, 24, n_nodes=50, seed=5, stress=1.0)
    w.load, w.cost = f32(w.load), f32(w.cost)
    oh = oracle_homes(w)
    iters = 5
    *_, tr = ro.solve_ADMM(oh, w.Rn, w.node_of, w.cost, w.kappa, iters, w.vset, w.vlow, w.vhigh,
                           mode="binary", keep=True, util_eps=1e-10)
    e = _engine(w, "binary")
    z = np.zeros_like(w.load)
    states = [(z, z, z)] + [(tr.P_est[k], tr.P_sch[k], tr.G[k]) for k in range(iters)]
    for k in range(iters):
        pe, ps, gm = (f32(a) for a in states[k])
        for t, a in ((e.P_est, pe), (e.P_sch, ps), (e.G, gm)):
            t.copy_(torch.from_numpy(np.ascontiguousarray(a[e.perm], np.float32)))
        e.op_cold = True
        e.step()
        pe_new = e.P_est.cpu().numpy()[e.inv_perm]
        assert np.abs(pe_new - tr.P_est[k]).max() < 1e-4, k
        P_sch, S, C = e.result()
        same = np.abs(S - tr.S[k]).max(axis=1) == 0
        assert same.mean() > 0.95, k
        obj_g = ro.home_objective(w.cost, oh, S.astype(float), pe, ps, gm, w.kappa)
        obj_r = ro.home_objective(w.cost, oh, tr.S[k], pe, ps, gm, w.kappa)
        assert np.max(np.abs(obj_g - obj_r) / np.maximum(1, np.abs(obj_r))) < 1e-4
        G = e.G.cpu().numpy()[e.inv_perm]
        assert np.abs(G - tr.G[k])[same].max() < 2e-3


def test_golden_feeder_admm_statistics(gpu_lib, golden, feeder_R):
    """15 iterations on the reference's 121144 feeder, com-2, 90% adoption, 4.8 kW.
    diff[1] is pinned exactly (tie-invariant); for k >= 2 Gurobi's arbitrary choice
    among exactly tied optima makes per-home values unreproducible, so the
    population statistics of the stored trajectory are compared instead."""
    from conftest import golden_homes
    from helpers import f32
    from revs_admm_amd.engine import AdmmEngine, pack_homes
    z, fd = golden
    oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
    n, T = oh.LOAD.shape
    e = AdmmEngine(f32(z["tariff_shift6"]), pack_homes(oh.ev, 4.8, 20.0, 0.2, 11, 23),
                   f32(oh.LOAD), np.arange(n), feeder_R, kappa=5.0, vset=1.03, vlow=0.95,
                   vhigh=1.05, mode="binary")
    diffs = e.run(15)
    ref = z["dis_a90_r4800_diff"]                       # (267, 15)
    mine = diffs[:, evi].T
    np.testing.assert_allclose(mine[:, 0], ref[:, 0], rtol=2e-6)
    m_ref, m_me = ref.mean(0), mine.mean(0)
    # (13 %: the faithful model reaches 5.2 - 10.7 % depending on the tie rule, two misreadings
    # of lpsolver.py 15.6 % and more: helpers.GOLDEN_BOUNDS, tests/test_oracle.py::
    # test_golden_distributed_trajectory_statistics; the full set of statistics on the GPU run:
    # tests/test_gpu_config4.py::test_golden_feeder_gpu_trajectory)
    from helpers import GOLDEN_BOUNDS
    assert np.abs(m_me / m_ref - 1).max() < GOLDEN_BOUNDS["mean"]
    P_sch, S, C = e.result()
    assert ((S[evi] > 0).sum(1) == 3).all() and (S[~oh.ev] == 0).all()
    np.testing.assert_allclose(C[evi][:, -1], 0.92, atol=1e-5)


def _nx_graph(fd, z):
    import networkx as nx
    g = nx.Graph()
    for nid, lab in zip(z["node_id"], fd.label):
        g.add_node(int(nid), label=lab.decode())
    for u, v, r in zip(fd.edge_u, fd.edge_v, fd.edge_r):
        g.add_edge(int(z["node_id"][u]), int(z["node_id"][v]), r=float(r))
    return g


def test_reference_call_surface(gpu_lib, golden):
    """lpsolver.solve_ADMM with the reference's own argument types (homes dict, networkx
    graph, tariff list) on the 121144 feeder; dict-shaped results; diff[1] == stored."""
    from revs_admm_amd.extract import get_homes_ev_param
    from revs_admm_amd.lpsolver import solve_ADMM
    z, fd = golden
    g = _nx_graph(fd, z)
    res = z["res_id"].tolist()
    ev = z["dis_a90_r4800_ev_homes"]
    homes = get_homes_ev_param({h: z["LOAD"][i].tolist() for i, h in enumerate(res)}, g, ev,
                               4800 * 1e-3, 20, 0.2, 11, 23)
    diff, P_sch, S, C = solve_ADMM(homes, g, z["tariff_shift6"].tolist(), "./gurobi", kappa=5.0,
                                   iter_max=2, vset=1.03, vlow=0.95, vhigh=1.05)
    assert sorted(diff) == [1, 2] and list(P_sch) == res and len(C[res[0]]) == 25
    got = np.array([diff[1][int(h)] for h in ev])
    np.testing.assert_allclose(got, z["dis_a90_r4800_diff"][:, 0], rtol=2e-6)
    for h in ev[:20]:
        assert abs(sum(S[int(h)]) - 3 * 4.8) < 1e-4 and abs(C[int(h)][-1] - 0.92) < 1e-5
        np.testing.assert_allclose(np.array(P_sch[int(h)]) - np.array(S[int(h)]),
                                   homes[int(h)]["LOAD"], atol=1e-5)


def test_individual_mode_golden(gpu_lib, golden):
    """revs_residence_solve (solve_residence, lpsolver.py:430-460) on the three stored
    individual-mode cases: same objective per residence as the reference's stored answer."""
    from conftest import golden_homes
    from oracle import revs_oracle as ro
    from revs_admm_amd.engine import pack_homes, residence_solve
    z, fd = golden
    for tag, rate in [("ind_a90_r4800", 4.8), ("ind_a70_r4800", 4.8), ("ind_a90_r3600", 3.6)]:
        oh, evi = golden_homes(z, tag, rate)
        p, soc, g = residence_solve(z["tariff_shift6"], pack_homes(oh.ev, rate, 20.0, 0.2, 11, 23),
                                    oh.LOAD)
        pref = np.zeros_like(oh.LOAD)
        pref[evi] = z[tag + "_P_ev"]
        o_gpu = ro.residence_objective(z["tariff_shift6"], oh, p.astype(float))
        o_ref = ro.residence_objective(z["tariff_shift6"], oh, pref)
        assert np.abs(o_gpu - o_ref).max() < 1e-6
        p_or, s_or, _ = ro.solve_residence(z["tariff_shift6"], oh)
        assert ((p > 0) == (p_or > 0)).all()                   # same tie rule as the oracle
        np.testing.assert_allclose(soc, s_or, atol=1e-6)


def test_process_group_path_on_one_gpu(gpu_lib):
    """The sharded code path (node-half hipGraphs + eager home pass + RCCL all-reduce) with a
    1-rank nccl group must give the same trajectory as the single-process path."""
    import os
    import torch
    import torch.distributed as dist
    from helpers import f32
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(800, 24, n_nodes=64, seed=13, binary_feasible=False, stress=0.9)
    w.load, w.cost = f32(w.load), f32(w.cost)
    from revs_admm_amd.engine import OperatorOptions
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        # dual Newton path, node-space ADMM fast path, general ADMM path
        for solver, fast in (("newton", True), ("admm", True), ("admm", False)):
            a = _engine(w, "relaxed_exact", op=OperatorOptions(solver=solver, node_fast=fast))
            da = a.run(5)
            b = _engine(w, "relaxed_exact", group=dist.group.WORLD,
                        op=OperatorOptions(solver=solver, node_fast=fast))
            db = b.run(5)
            if solver == "newton":
                # (Newton iterations and evaluations of every solve; pivots where the one-GPU run read them:
                # an iteration finished inside the folded chain, second Newton step included, books -1)
                assert set(b.op_path_hist) == {"dual"}
                assert [h[:2] for h in a.newton_hist] == [h[:2] for h in b.newton_hist]
                assert all(x[2] == y[2] for x, y in zip(a.newton_hist, b.newton_hist) if x[2] >= 0)
                assert max(n for n, _, _ in b.newton_hist) >= 1   # rows bind on the way
            else:
                assert max(b.op_iters_hist) >= 25                 # rows bind on the way
                assert set(b.op_path_hist) == {"home"} if not fast else "node" in b.op_path_hist
            if solver == "admm" and not fast:
                assert isinstance(b._graph, list) and len(b._graph) == 2   # sharded graphs ran
            assert a.op_iters_hist == b.op_iters_hist
            np.testing.assert_allclose(db, da, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(b.result()[1], a.result()[1], atol=1e-5)
            assert b.residuals(1e-4)[:3] == pytest.approx(a.residuals(1e-4)[:3], rel=1e-5)
        # long enough for the steady state: the native two-phase iteration around the all-reduce
        # of p (home pass folded into the sweep, kept and discarded sweeps) == the one-GPU run
        w2 = make_workload(3000, 24, n_nodes=100, seed=5, binary_feasible=False, stress=1.02)
        # (round 4: with the library's communicator the binding steady state runs through the folded chain
        # sharded too -- one all-reduce of both folded sum arrays per iteration)
        a = _engine(w2, "pdhg")
        b = _engine(w2, "pdhg", group=dist.group.WORLD)
        da, db = a.run(25), b.run(25)
        assert b._plan is not None and b.spec_hist == a.spec_hist and a.spec_hist[0] > 0 < a.spec_hist[1]
        np.testing.assert_array_equal(db, da)
        np.testing.assert_array_equal(b.result()[0], a.result()[0])
        # the streaming steady state with the library's OWN communicator in the loop: sweep ->
        # ncclAllReduce of the node sums on the compute stream -> sweep, launches made in chunks,
        # no host read in between (revs_plan_stream_run) == the one-GPU run, bit for bit
        a = _engine(w2, "pdhg", op=OperatorOptions(stream_block_single=False))   # every launch judges itself
        b = _engine(w2, "pdhg", group=dist.group.WORLD)
        assert b._comm is not None and b._tree is not None
        for chunk in (3, 40, 37):
            a.run_steps(chunk)
            b.run_steps(chunk)
        assert a.spec_hist == b.spec_hist and a.spec_hist[0] > 40 and a.spec_hist[1] > 0
        for name in ("P_est", "P_sch", "G", "diff"):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert b.residuals(1e-4)[:3] == pytest.approx(a.residuals(1e-4)[:3], rel=1e-6)
        # (sharded, the verdicts are taken by blocks: one all-reduce per block of sweeps)
        assert b._block == 32 and a._block == 0 and a.stream_calls == b.stream_calls
        for opt in (OperatorOptions(stream_block=1), OperatorOptions(stream_overlap=False)):
            c = _engine(w2, "pdhg", group=dist.group.WORLD, op=opt)
            for chunk in (3, 40, 37):
                c.run_steps(chunk)
            assert c._block == (0 if opt.stream_block == 1 else 32)  # (0: one all-reduce per sweep)
            assert c.stream_calls == a.stream_calls
            for name in ("P_est", "P_sch", "G", "diff"):
                assert torch.equal(getattr(a, name), getattr(c, name)), name
    finally:
        dist.destroy_process_group()


def test_centralized_mode_golden(gpu_lib, golden):
    """solve_central (lpsolver.py:463-502) on the 121144 feeder == the reference's stored
    centralized result (all chargers off, SOC 0.2, P_res = LOAD); an overloaded feeder raises."""
    from revs_admm_amd import _lib
    from revs_admm_amd.extract import get_homes_ev_param
    from revs_admm_amd.lpsolver import solve_central
    z, fd = golden
    g = _nx_graph(fd, z)
    res = z["res_id"].tolist()
    ev = z["cen_a90_r4800_ev_homes"]
    homes = get_homes_ev_param({h: z["LOAD"][i].tolist() for i, h in enumerate(res)}, g, ev,
                               4.8, 20, 0.2, 11, 23)
    p, s, gg = solve_central(z["tariff_shift6"].tolist(), homes, g, None, 1.03, 0.90, 1.05)
    np.testing.assert_allclose(np.array([p[int(h)] for h in ev]), z["cen_a90_r4800_P_ev"], atol=1e-12)
    np.testing.assert_allclose(np.array([s[int(h)] for h in ev]), z["cen_a90_r4800_SOC"], atol=1e-7)
    np.testing.assert_allclose(np.array([gg[h] for h in res]), z["cen_a90_r4800_P_res"], atol=1e-9)
    with pytest.raises(_lib.RevsError, match="No solution found"):
        solve_central(z["tariff_shift6"].tolist(), homes, g, None, 1.03, 0.99, 1.05)


def test_centralized_mode_negative_prices(gpu_lib, golden):
    """solve_central where some prices are negative: the chargers go on in the most negative slots of
    their windows, as many as the SOC box allows, as long as the voltage rows stay respected
    (checked on the GPU) -- the optimum of the reference's model (lpsolver.py:463-502), which the
    oracle solves as a MILP.  When the rows would bind, it says so instead of returning something."""
    from oracle import revs_oracle as ro
    from revs_admm_amd.extract import get_homes_ev_param
    from revs_admm_amd.lpsolver import compute_Rmat, homes_to_arrays, solve_central
    z, fd = golden
    g = _nx_graph(fd, z)
    res = z["res_id"].tolist()
    ev = z["cen_a90_r4800_ev_homes"]
    homes = get_homes_ev_param({h: z["LOAD"][i].tolist() for i, h in enumerate(res)}, g, ev,
                               4.8, 20, 0.2, 11, 23)
    tariff = np.array(z["tariff_shift6"], float)
    tariff[[12, 13, 20]] = [-0.02, -0.05, -0.02]          # a tie between slots 12 and 20
    tariff[3] = -1.0                                      # outside every window: never used
    p, s, gg = solve_central(tariff.tolist(), homes, g, None, 1.03, 0.90, 1.05)
    P = np.array([p[h] for h in res])
    assert (P[:, 3] == 0).all() and P[:, 13].max() == pytest.approx(4.8) and P.sum() > 0
    nonsub = [n for n in g.nodes if g.nodes[n]["label"] != "S"]
    ri = [nonsub.index(n) for n in res]
    R_res = compute_Rmat(g)[np.ix_(ri, ri)]
    load_, rec = homes_to_arrays(homes, res)
    p_ref, g_ref, tot_ref = ro.solve_central_ref(tariff, ro.homes_from_records(load_, rec), R_res, 1.03, 0.90, 1.05)
    tot = sum(float(np.dot(tariff, gg[h])) for h in res)
    assert tot == pytest.approx(tot_ref, rel=1e-12, abs=1e-9)
    assert P.sum(1) == pytest.approx(p_ref.sum(1))            # same number of slots per residence
    S = np.array([s[h] for h in res])
    assert S.max() <= 1.0 + 1e-9 and (np.diff(S, axis=1) >= -1e-12).all()
    # a floor the extra charging cannot respect: the rows bind, the residences do not decouple
    v = -(R_res @ (load_ + P))
    vmin_tight = float(np.sqrt(1.03 ** 2 + 0.5 * (v.min() + (-(R_res @ load_)).min())))
    with pytest.raises(NotImplementedError, match="voltage rows bind"):
        solve_central(tariff.tolist(), homes, g, None, 1.03, vmin_tight, 1.05)


def _operator_kkt_f64(e):
    """KKT certificate of the operator's last answer on the node-space fast path, recomputed
    in float64 on the host from the engine's state: stationarity kappa d + Rs y = 0, rows
    inside their (sqrt(n_m)-scaled) bounds, y only on active rows with the right sign.
    Returns the number of active upper rows."""
    Q, lam = e.Q.cpu().numpy(), e.s.cpu().numpy()
    Rs = (Q * lam[None, :]) @ Q.T
    p0, d, y = e.p0.cpu().numpy(), e.dnode.cpu().numpy(), e.yv.cpu().numpy()
    sq = e.sqrt_n.cpu().numpy()[:, None]
    v = Rs @ (p0 + d)
    scale = max(abs(e.vlo), abs(e.vhi)) * sq.max()
    assert (v <= sq * e.vhi + 1e-7 * scale).all() and (v >= sq * e.vlo - 1e-7 * scale).all()
    assert np.abs(e.kappa * d + Rs @ y).max() < 1e-6 * e.kappa * max(1.0, np.abs(p0).max())
    act_hi = v >= sq * e.vhi - 1e-6 * scale
    act_lo = v <= sq * e.vlo + 1e-6 * scale
    ytol = 1e-6 * max(np.abs(y).max(), 1e-300)
    assert (np.abs(y[~(act_hi | act_lo)]) <= ytol).all() and (y[act_hi] >= -ytol).all() \
        and (y[act_lo] <= ytol).all()
    return int(act_hi.sum())


def _dual_kkt_f64(e, w, pe0, ps0, gm0):
    """KKT certificate of the dual Newton path's last answer, recomputed in float64 on the
    host from the multipliers y and the state the operator was given: stationarity
    g = max(g0 - (R^T y / kappa)[node], 0) is the answer handed to the homes, the rows of
    R.(A g) respect their bounds, y is non-zero only on rows at a bound, with the right
    sign.  Returns the number of rows at the upper bound."""
    y = e.yd[0].cpu().numpy()
    node = w.node_of[e.perm]
    g0 = 0.5 * (pe0.astype(np.float64) + ps0) - gm0.astype(np.float64) / e.kappa
    g = np.maximum(g0 - (w.Rn.T @ y / e.kappa)[node], 0.0)
    got = e.P_est.cpu().numpy().astype(np.float64)       # after the swap: the operator's answer
    assert np.abs(got - g).max() < 1e-6 * max(1.0, np.abs(g).max())
    p = np.zeros((e.M, e.T))
    np.add.at(p, node, g)
    v = w.Rn @ p
    scale = max(abs(e.vlo), abs(e.vhi))
    assert v.max() <= e.vhi + 1e-7 * scale and v.min() >= e.vlo - 1e-7 * scale
    act_hi, act_lo = v >= e.vhi - 1e-6 * scale, v <= e.vlo + 1e-6 * scale
    assert (y[~(act_hi | act_lo)] == 0).all() and (y[act_hi] >= 0).all() and (y[act_lo] <= 0).all()
    return int(act_hi.sum())


def test_full_size_invariants(gpu_lib):
    """BASELINE size (100k homes x T=24, 2048 nodes): properties that need no oracle run.
    SOC rows, windows and energy bounds hold for every home; the epilogue identities
    P_sch = LOAD + S, G += kappa/2 (P_est - P_sch), diff = |P_est - P_sch|/T hold; the
    operator's answer is voltage-feasible and non-negative; halves of the homes solved
    separately with the same P_est give the same schedules (sharding invariance)."""
    import torch
    from revs_admm_amd.engine import AdmmEngine
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.1)
    e = _engine(w, "pdhg")
    n_active = []
    for _ in range(3):
        G_before = e.G.clone()
        pe0, ps0 = e.P_est.cpu().numpy(), e.P_sch.cpu().numpy()
        e.step()
        assert e.op_path_hist[-1] == "dual"
        n_active.append(_dual_kkt_f64(e, w, pe0, ps0, G_before.cpu().numpy()))
    assert max(n_active) > 0                        # the voltage rows do bind on the way
    # the same state through the ADMM forms' node-space fast path carries its own certificate
    from revs_admm_amd.engine import OperatorOptions
    ea = _engine(w, "pdhg", op=OperatorOptions(solver="admm"))
    for _ in range(2):
        ea.step()
        if ea.op_path_hist[-1] == "node":
            _operator_kkt_f64(ea)
    P_sch, S, C = e.result()
    h = w.homes
    ev = h["ev"] == 1
    t = np.arange(24)[None, :]
    win = ev[:, None] & (t >= h["start"][:, None]) & (t < h["end"][:, None])
    assert (S[~win] == 0).all() and (S >= 0).all() and (S <= h["rating"][:, None] * (1 + 1e-6)).all()
    np.testing.assert_allclose(P_sch, w.load.astype(np.float32) + S, rtol=1e-6, atol=1e-6)
    soc = np.where(ev[:, None], h["initial"][:, None] + np.cumsum(S, 1) / h["capacity"][:, None], 0)
    np.testing.assert_allclose(C[:, 1:], soc, atol=2e-5)
    assert (C[ev, -1] >= 0.9 - 2e-4).all() and (C <= 1 + 2e-4).all() and (np.diff(C, axis=1) >= -1e-6).all()
    pe = e.P_est.cpu().numpy()[e.inv_perm]
    chk = pe - P_sch
    np.testing.assert_allclose(e.diff.cpu().numpy()[e.inv_perm], np.linalg.norm(chk, axis=1) / 24,
                               rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(e.G.cpu().numpy()[e.inv_perm],
                               G_before.cpu().numpy()[e.inv_perm] + 2.5 * chk, rtol=1e-4, atol=1e-4)
    v = e.voltage(e.P_est).cpu().numpy()
    assert pe.min() >= 0 and v.max() <= e.vhi * (1 + 1e-4)
    assert e.residuals(1e-4)[2] == pytest.approx(np.max(np.linalg.norm(chk, axis=1) / 24), rel=1e-4)



@pytest.mark.parametrize("solver", ["newton", "admm"])
def test_relaxed_trajectory_T96(gpu_lib, solver):
    """15-minute slots (BASELINE config 4/5 shape): T = 96 uses the 16-lane x 6-slot home groups, 96
    independent slot problems in the dual Newton path and the 192-column concatenated product
    in the ADMM forms; trajectory vs oracle as for T = 24."""
    from helpers import f32, oracle_homes
    from oracle import revs_oracle as ro
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(300, 96, n_nodes=40, seed=17, binary_feasible=False, stress=0.9)
    w.load, w.cost = f32(w.load), f32(w.cost)
    from revs_admm_amd.engine import OperatorOptions
    e = _engine(w, "pdhg", op=OperatorOptions(solver=solver))
    assert e.T == 96 and (solver == "newton" or e.cat)
    diffs = e.run(5)
    assert set(e.op_path_hist) == ({"dual"} if solver == "newton" else {"node", "home"} & set(e.op_path_hist))
    P_sch, S, C = e.result()
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oracle_homes(w), w.Rn, w.node_of, w.cost, w.kappa, 5,
                                               w.vset, w.vlow, w.vhigh, mode="relaxed",
                                               util_eps=1e-10)
    print(f"T = 96 trajectory, {solver}: |S - oracle| {np.abs(S - S_ref).max():.2e} kW")
    assert np.abs(diffs - d_ref).max() < 1e-3 * max(1.0, d_ref.max())
    assert np.abs(S - S_ref).max() < 2e-5 and np.abs(C - C_ref).max() < 5e-6


def test_config3_all_communities_90pct_T96(gpu_lib, golden, feeder_R):
    """BASELINE config 3: 121144 feeder, ALL communities, 90 % adoption (1013 EVs, drawn as
    revs_fixture.py:174-177 does), 4.8 kW, T = 96 -- hourly load and tariff held over four
    15-minute slots, window 44..92.  Every slot has binding voltage rows (up to ~50 multipliers
    per slot); three ADMM iterations of the relaxed problem vs the oracle, all operator solves
    on the dual Newton path (tests/tools/feeder_config3.py times the 15-iteration run)."""
    from helpers import f32
    from oracle import revs_oracle as ro
    from revs_admm_amd.engine import AdmmEngine, pack_homes
    z, fd = golden
    res_ids = z["res_id"]
    n = len(res_ids)
    np.random.seed(1234)
    ev_homes = np.random.choice(res_ids, int(90 * 1e-2 * n), replace=False)
    idx = {h: i for i, h in enumerate(res_ids)}
    ev = np.zeros(n, bool)
    ev[[idx[h] for h in ev_homes]] = True
    assert ev.sum() == 1013
    LOAD = f32(np.repeat(z["LOAD"], 4, axis=1))
    cost = f32(np.repeat(z["tariff_shift6"], 4))
    oh = ro.Homes.uniform(LOAD, ev, 4.8, 20.0, 0.2, 44, 92)
    e = AdmmEngine(cost, pack_homes(ev, 4.8, 20.0, 0.2, 44, 92), LOAD, np.arange(n), feeder_R,
                   kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode="relaxed_exact")
    # (round 4: the engine gets the matrix only, as the reference's Utility does, recovers the radial feeder from it
    # -- 1126 residence rows, junctions without a residence as extra tree nodes -- and judges every row by the tree
    # form of R p: no dense product on this feeder)
    assert e._tree is not None and e._tree_newton and 1126 <= e._tree.n <= 1696
    diffs = e.run(3)
    assert set(e.op_path_hist) == {"dual"}
    y = e.yd[0].cpu().numpy()
    assert 8 < (y != 0).sum(0 if y.shape[0] == n else 1).max() <= 128
    P_sch, S, C = e.result()
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oh, feeder_R, np.arange(n), cost, 5.0, 3, 1.03,
                                               0.95, 1.05, mode="relaxed", util_eps=1e-10)
    print(f"config 3, closed form, 3 iterations: |S - oracle| {np.abs(S - S_ref).max():.2e} kW")
    assert np.abs(diffs - d_ref).max() < 1e-3 * max(1.0, d_ref.max())
    assert np.abs(S - S_ref).max() < 2e-5 and np.abs(C - C_ref).max() < 5e-6


def _config3(golden):
    """BASELINE config 3's inputs (all communities, 90 % adoption drawn as revs_fixture.py:174-177 does, T = 96)."""
    from helpers import f32
    from oracle import revs_oracle as ro
    from revs_admm_amd.engine import pack_homes
    z, fd = golden
    res_ids = z["res_id"]
    n = len(res_ids)
    np.random.seed(1234)
    ev_homes = np.random.choice(res_ids, int(90 * 1e-2 * n), replace=False)
    idx = {h: i for i, h in enumerate(res_ids)}
    ev = np.zeros(n, bool)
    ev[[idx[h] for h in ev_homes]] = True
    LOAD = f32(np.repeat(z["LOAD"], 4, axis=1))
    cost = f32(np.repeat(z["tariff_shift6"], 4))
    return n, cost, LOAD, ro.Homes.uniform(LOAD, ev, 4.8, 20.0, 0.2, 44, 92), pack_homes(ev, 4.8, 20.0, 0.2, 44, 92)


def test_config3_pdhg_15_iterations(gpu_lib, golden, feeder_R):
    """Config 3 as bench.py's value_feeder_121144 runs it -- AdmmEngine.run(15), the reference's iter_max -- with PDHG
    residences (the relaxed QP has unique optima: the whole trajectory is comparable): every residence's diff at all
    15 iterations, the final schedules and SOC against the oracle's run."""
    from oracle import revs_oracle as ro
    from revs_admm_amd.engine import AdmmEngine
    n, cost, LOAD, oh, rec = _config3(golden)
    e = AdmmEngine(cost, rec, LOAD, np.arange(n), feeder_R, kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode="pdhg")
    diffs = e.run(15)
    assert set(e.op_path_hist) == {"dual"} and e._tree_eval
    P_sch, S, C = e.result()
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oh, feeder_R, np.arange(n), cost, 5.0, 15, 1.03, 0.95, 1.05,
                                               mode="relaxed", util_method="dual")
    print(f"config 3, PDHG, 15 iterations: max |diff - oracle| {np.abs(diffs - d_ref).max():.2e} (diff max {d_ref.max():.2e}, "
          f"last {d_ref[-1].max():.2e}), |S - oracle| {np.abs(S - S_ref).max():.2e} kW, operator evaluations {sum(e.op_iters_hist)}")
    assert np.abs(diffs - d_ref).max() < 1e-3 * max(1.0, d_ref.max())
    assert np.abs(diffs[-1] - d_ref[-1]).max() < 0.02 * d_ref[-1].max() + 2e-6
    assert np.abs(S - S_ref).max() < 5e-5 and np.abs(P_sch - P_ref).max() < 5e-5 and np.abs(C - C_ref).max() < 2e-5
    assert getattr(e, "polish_unsettled", 0) == 0          # (status bit 2: no residence was left at PDHG's own tolerance)


def test_config3_binary_teacher_forced_15_iterations(gpu_lib, golden, feeder_R):
    """Config 3 with the reference's on/off chargers, 15 iterations, teacher-forced (every iteration starts from the
    ORACLE's state rounded to float: one flipped exact tie would otherwise change every later iterate; the engine keeps
    its own multipliers and fast-path flags), residences ranking in double (revs_pdhg_t::keys64 = 1).  Per iteration:
    the operator's answer against the oracle's, the on/off pattern against the oracle's home solve of the SAME float
    state -- identical for every residence -- and the dual update."""
    from helpers import f32
    from oracle import revs_oracle as ro
    from revs_admm_amd.engine import AdmmEngine
    n, cost, LOAD, oh, rec = _config3(golden)
    iters = 15
    *_, tr = ro.solve_ADMM(oh, feeder_R, np.arange(n), cost, 5.0, iters, 1.03, 0.95, 1.05, mode="binary", keep=True,
                           util_method="dual")
    e = AdmmEngine(cost, rec, LOAD, np.arange(n), feeder_R, kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode="binary",
                   pdhg=dict(keys64=1))
    z0 = np.zeros_like(LOAD)
    states = [(z0, z0, z0)] + [(tr.P_est[k], tr.P_sch[k], tr.G[k]) for k in range(iters)]
    worst_pe = 0.0
    for k in range(iters):
        pe, ps, gm = (f32(a) for a in states[k])
        chain_ok = e._chain_ok
        e.set_state(pe, ps, gm)
        e._chain_ok = chain_ok
        e.step()
        pe_new = e.P_est.cpu().numpy()[e.inv_perm]
        worst_pe = max(worst_pe, float(np.abs(pe_new - tr.P_est[k]).max()))
        P_sch, S, C = e.result()
        p_chk = ro.home_solve_binary(cost, oh, pe, ps, gm, 5.0)[0]
        assert ((S > 0) == (p_chk > 0)).all(), (k, int(((S > 0) != (p_chk > 0)).any(axis=1).sum()))
        G = e.G.cpu().numpy()[e.inv_perm]
        np.testing.assert_allclose(G, gm + 2.5 * (pe_new.astype(np.float64) - (p_chk + LOAD)), rtol=1e-5, atol=2e-5)
    print(f"config 3, on/off chargers, teacher-forced x{iters}: worst |P_est - oracle| {worst_pe:.2e} kW, Newton iterations "
          f"{[h[0] for h in e.newton_hist]}")
    assert worst_pe < 1e-4


def test_config0_com2_30pct_adoption(gpu_lib, golden, feeder_R):
    """BASELINE config 0: 121144 feeder, community 2, 30 % adoption, 4.8 kW, T = 24 (the
    reference's test-optimizer.py case; no stored result exists for it).  EV homes are drawn
    as revs_fixture.py:174-177 does; iteration 1 must equal the oracle exactly and iteration
    2's operator answer must equal the oracle's QP solution."""
    from helpers import f32
    from oracle import revs_oracle as ro
    from revs_admm_amd.engine import AdmmEngine, pack_homes
    z, fd = golden
    res_ids = z["res_id"]
    com2 = z["com_flat"][z["com_offsets"][1]:z["com_offsets"][2]]
    np.random.seed(1234)
    ev_homes = np.random.choice(com2, int(30 * 1e-2 * len(com2)), replace=False)
    idx = {h: i for i, h in enumerate(res_ids)}
    ev = np.zeros(len(res_ids), bool)
    ev[[idx[h] for h in ev_homes]] = True
    assert ev.sum() == 89
    oh = ro.Homes.uniform(f32(z["LOAD"]), ev, 4.8, 20.0, 0.2, 11, 23)
    cost = f32(z["tariff_shift6"])
    n, T = oh.LOAD.shape
    e = AdmmEngine(cost, pack_homes(ev, 4.8, 20.0, 0.2, 11, 23), oh.LOAD, np.arange(n), feeder_R,
                   kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode="binary")
    d = e.run(2)
    zero = np.zeros((n, T))
    p1, s1, g1, st = ro.home_solve_binary(cost, oh, zero, zero, zero, 5.0)
    np.testing.assert_allclose(d[0], np.linalg.norm(g1, axis=1) / T, rtol=2e-6)
    # iteration 2: operator projects g0 = P_sch[1]; homes repeat iteration 1's problem
    g1f = f32(g1)
    vlo, vhi = ro.voltage_limits(1.03, 0.95, 1.05)
    ref = ro.utility_solve(feeder_R, np.arange(n), ro.utility_g0(zero, g1f, f32(-2.5 * g1f), 5.0),
                           5.0, vlo, vhi, eps=1e-10)
    pe2 = e.P_est.cpu().numpy()[e.inv_perm].astype(np.float64)
    assert np.abs(pe2 - ref).max() < 5e-5
    P_sch, S, C = e.result()
    assert (np.abs(S - p1) < 1e-6).all()             # same home problem as iteration 1
    np.testing.assert_allclose(d[1], np.linalg.norm(ref - g1, axis=1) / T, atol=2e-5)
    assert e.op_path_hist[-1] == "dual" and pe2.min() == 0.0   # residences clamped at zero


def test_config0_15_iterations(gpu_lib, golden, feeder_R):
    """BASELINE config 0 for the reference's 15 iterations (lpsolver.py:243, on/off chargers) against the oracle's own
    run of it.  Per residence the runs part ways at the first exactly tied MIQP optimum that float state and double
    state resolve differently, so beyond iteration 1 they are compared on the tie-robust
    statistics the stored 90 % run is pinned with: mean diff[k] of the EV residences per iteration, the sorted-diff
    distance, EV slots per tariff block of the final schedules and the total EV energy cost."""
    from helpers import f32
    from oracle import revs_oracle as ro
    from revs_admm_amd.engine import AdmmEngine, pack_homes
    z, fd = golden
    res_ids = z["res_id"]
    com2 = z["com_flat"][z["com_offsets"][1]:z["com_offsets"][2]]
    np.random.seed(1234)
    ev_homes = np.random.choice(com2, int(30 * 1e-2 * len(com2)), replace=False)
    idx = {h: i for i, h in enumerate(res_ids)}
    evi = np.array([idx[h] for h in ev_homes])
    ev = np.zeros(len(res_ids), bool)
    ev[evi] = True
    oh = ro.Homes.uniform(f32(z["LOAD"]), ev, 4.8, 20.0, 0.2, 11, 23)
    cost = f32(z["tariff_shift6"])
    n, T = oh.LOAD.shape
    e = AdmmEngine(cost, pack_homes(ev, 4.8, 20.0, 0.2, 11, 23), oh.LOAD, np.arange(n), feeder_R,
                   kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode="binary")
    d = e.run(15)
    P_sch, S, C = e.result()
    d_or, P_or, S_or, C_or = ro.solve_ADMM(oh, feeder_R, np.arange(n), cost, 5.0, 15, 1.03, 0.95, 1.05, mode="binary",
                                           util_method="dual")
    rel = np.abs(d[:, evi].mean(1) / d_or[:, evi].mean(1) - 1)
    w1 = max(np.abs(np.sort(d[k, evi]) - np.sort(d_or[k, evi])).mean() / d_or[k, evi].mean() for k in range(1, 15))
    on, on_or = S[evi] > 1e-6, S_or[evi] > 1e-6
    blocks = max(abs(int(on[:, cost == b].sum()) - int(on_or[:, cost == b].sum())) for b in np.unique(cost))
    ev_cost = abs((S[evi] * cost).sum() / (S_or[evi] * cost).sum() - 1)
    same = float((np.abs(S - S_or).max(1) == 0).mean())
    print(f"config 0, 15 iterations: mean diff[k] GPU/oracle - 1 {np.round(rel, 5).tolist()}, sorted-diff distance {w1:.4f}, "
          f"EV slots per tariff block off by {blocks}, EV cost off by {ev_cost:.2e}, identical final schedules {same:.3f}")
    # (bounds: those of the stored run's pin, helpers.GOLDEN_BOUNDS -- the spread between two consistent tie rules)
    assert rel[0] < 1e-5 and rel.max() < 0.13 and w1 < 0.145 and blocks <= 6 and ev_cost < 3.2e-3


def test_revs_fixture_end_to_end(gpu_lib, golden, tmp_path):
    """The reference's driver flow (test-optimizer.py): REVS(**file_params).read_inputs(...)
    -> get_individual_optimal / get_distributed_optimal / get_centralized_optimal(save=True),
    on input files written in the reference's formats (tariff txt, community txt, home-load
    csv in W with hour1..hour24 columns, pickled networkx feeder)."""
    import pickle
    from revs_admm_amd.revs_fixture import REVS
    z, fd = golden
    data = tmp_path / "input"
    data.mkdir()
    res = z["res_id"].tolist()
    (data / "DVP-tariff.txt").write_text(" ".join(repr(float(x)) for x in z["tariff_raw"]))
    offs = z["com_offsets"]
    (data / "121144-com.txt").write_text("\n".join(
        " ".join(str(int(h)) for h in z["com_flat"][offs[i]:offs[i + 1]]) for i in range(5)))
    unshift = np.roll(z["LOAD"], 6, axis=1) * 1e3               # file is unshifted, in W
    with open(data / "121-home-load.csv", "w") as f:
        f.write("hid," + ",".join(f"hour{i + 1}" for i in range(24)) + "\n")
        for h, row in zip(res, unshift):
            f.write(str(h) + "," + ",".join(repr(float(v)) for v in row) + "\n")
    with open(data / "121144-dist-net.gpickle", "wb") as f:
        pickle.dump(_nx_graph(fd, z), f)
    fx = REVS(networkID=121144, regionID=121, comunityID=2, tariffID="DVP",
              optimizer_mode="distributed", data_path=str(data), out_path=str(tmp_path / "out"),
              fig_path=str(tmp_path / "figs"))
    inp = dict(adoption=90, rating=4800, seed=1234, capacity=20, initial_soc=0.2, start_time=11,
               end_time=23, shift_time=6)
    tariff, homes, dist, save = fx.read_inputs(**inp)
    np.testing.assert_allclose(tariff, z["tariff_shift6"])
    assert list(save["ev_homes"]) == z["dis_a90_r4800_ev_homes"].tolist()
    np.testing.assert_allclose(homes[res[5]]["LOAD"], z["LOAD"][5], rtol=1e-12)
    opt = dict(v0=1.03, vmin=0.90, vmax=1.05, max_iterations=3, kappa=5.0, **save, **inp)
    Pres, Pev, soc = fx.get_distributed_optimal(tariff, homes, dist, save=True, **opt)
    out = tmp_path / "out" / "121144-com2" / "distributed" / "adopt90-rating4800-seed1234.txt"
    txt = out.read_text()
    assert sum(l.startswith("####") for l in txt.split("\n")) == 8
    assert "EV Convergence over Iterations" in txt
    ev0 = int(save["ev_homes"][0])
    line = [l for l in txt.split("\n") if l.startswith(f"{ev0}:\t")][-1]        # diff row
    d1 = float(line.split("\t")[1].split(" ")[0])
    assert abs(d1 - z["dis_a90_r4800_diff"][0, 0]) < 2e-6 * d1
    assert abs(sum(Pev[ev0]) - 14.4) < 1e-4 and abs(soc[ev0][-1] - 0.92) < 1e-5
    # the other two modes of the fixture
    fx.optim = "individual"
    Pres_i, Pev_i, soc_i = fx.get_individual_optimal(tariff, homes, **opt)
    assert abs(sum(Pev_i[ev0]) - 14.4) < 1e-4
    Pres_c, Pev_c, soc_c = fx.get_centralized_optimal(tariff, homes, dist, **opt)
    assert sum(Pev_c[ev0]) == 0.0 and soc_c[ev0][-1] == pytest.approx(0.2)


@pytest.mark.parametrize("mode,T", [("pdhg", 24), ("binary", 24), ("pdhg", 96), ("relaxed_exact", 12)])
def test_native_step_and_fused_home_pass_change_nothing(gpu_lib, mode, T, monkeypatch):
    """The steady-state iteration as one native call (revs_plan_spec_step), with and without
    the next evaluation's home pass folded into the sweep, against the Python-issued
    iteration: same speculation history, same schedules and residuals bit for bit (the
    fused node sums are accumulated with atomics, but with y = 0 nothing depends on their
    last bits), over a run with kept AND discarded speculative sweeps; also with the sweep
    recomputing P_est[k+1] from the state instead of loading it."""
    from revs_admm_amd.engine import OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    # (binary schedules at stress 1 keep the rows moving: no steady state; at 0.5 the rows stay slack)
    # T = 96: 16-lane x 6-slot home groups, product and rows as two kernels (the one-launch form is for
    # T <= 32); T = 12: half-empty lane groups
    w = make_workload(3000 if T <= 24 else 1500, T, n_nodes=100, seed=5,
                      binary_feasible=(mode == "binary"), stress=0.5 if mode == "binary" else 1.02)
    runs = []
    # (rec: the sweep recomputes the operator's steady-state answer instead of reading it --
    # the rule for large problems, forced here)
    for plan, fuse, rec in ((True, True, True), (True, True, False), (True, False, False), (False, False, False)):
        e = _engine(w, mode, op=OperatorOptions(fuse_home_pass=fuse, native_plan=plan, recompute_pe_new=rec))
        assert (e._plan is not None) == plan and e.recompute_pe_new == (rec and plan)
        d = e.run(25)
        runs.append((d, e.result(), e.P_est.cpu().numpy(), list(e.spec_hist),
                     [h[0] for h in e.newton_hist]))
    ref = runs[0]
    assert ref[3][0] > 0 and (mode == "binary" or T != 24 or ref[3][1] > 0)
    for i, r in enumerate(runs[1:]):
        assert r[3] == ref[3] and r[4] == ref[4]
        # the native forms among themselves: bit for bit.  The Python-issued iteration (last run: no
        # plan) judges the Newton evaluations' rows by the dense product where the plan uses the tree
        # form of R p (round 3): the same decisions, values to the last bits of a float
        exact = i < 2
        cmp = np.testing.assert_array_equal if exact else (lambda x, y: np.testing.assert_allclose(x, y, rtol=0, atol=2e-5))
        cmp(r[0], ref[0])
        for a, b in zip(r[1], ref[1]):
            cmp(a, b)
        cmp(r[2], ref[2])


@pytest.mark.parametrize("mode,stress,redo", [("binary", 1.0, 2), ("binary", 1.0, 0), ("relaxed_exact", 1.3, 2), ("pdhg", 1.3, 2)])
def test_chained_newton_iteration_changes_nothing(gpu_lib, mode, stress, redo):
    """The binding steady state enqueued whole (evaluation, small model, step decided on the
    device, evaluation, sweep: engine._chain_launch) against the driver that reads every
    evaluation before going on: same schedules and multipliers, bit for bit -- PDHG homes included."""
    from helpers import f32
    from revs_admm_amd.engine import OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(6000, 24, n_nodes=150, seed=5, binary_feasible=(mode == "binary"), stress=stress)
    w.load, w.cost = f32(w.load), f32(w.cost)
    runs = []
    for chain in (True, False):
        # (redo = 0: a trial that needs a second Newton step is handed back at the step and the solve goes on
        # in the caller's loop; 2: the second and third steps are taken inside the native call)
        e = _engine(w, mode, op=OperatorOptions(chain=chain, fold_redo=redo))
        d = e.run(60)
        runs.append((d, e.result(), e.yd[0].cpu().numpy(), e))
    (d1, r1, y1, e1), (d0, r0, y0, e0) = runs
    assert e1.chain_hist[0] > 5 and e0.chain_hist == [0, 0], (e1.chain_hist, e1.newton_hist[-20:])
    if mode == "binary":
        # (on/off chargers: some iterations need a second Newton step -- the folded chain hands its own
        # step back, revs_chain_fold_state_t::resume = 2, and the solve goes on from it)
        assert e1.fold_steps > 0 and e0.fold_steps == 0
    # (PDHG homes too since round 4: the folded chain's sweeps write the carried multipliers to a spare array, so a
    # rejected sweep leaves the warm start of the sweep that replaces it untouched)
    np.testing.assert_array_equal(d1, d0)
    for a, b in zip(r1, r0):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(y1, y0)


@pytest.mark.parametrize("mode,stress", [("relaxed_exact", 1.0), ("pdhg", 1.02), ("binary", 0.5), ("binary", 1.0),
                                         ("relaxed_exact", 1.3)])
def test_run_steps_equals_repeated_step(gpu_lib, mode, stress):
    """run_steps (consecutive steady-state iterations inside one native call, buffer rotation
    included: revs_plan_spec_run, and revs_plan_chain_run where rows keep binding; discarded
    sweeps and everything else through step()) leaves
    the state that the same number of step() calls leaves, bit for bit -- through the
    transient, kept and discarded speculative sweeps, and in chunks of any size."""
    from helpers import f32
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(8000, 24, n_nodes=200, seed=3, binary_feasible=(mode == "binary"), stress=stress)
    w.load, w.cost = f32(w.load), f32(w.cost)
    a, b = _engine(w, mode), _engine(w, mode)
    for _ in range(90):
        a.step(write_sc=False)
    for chunk in (1, 7, 30, 2, 50):
        b.run_steps(chunk)
    assert a.iteration == b.iteration == 90
    assert a.spec_hist == b.spec_hist and a.chain_hist == b.chain_hist, (a.spec_hist, b.spec_hist)
    assert a.spec_hist[0] > 20 or a.chain_hist[0] > 20, (a.spec_hist, a.chain_hist)
    assert a.op_iters_hist == b.op_iters_hist
    assert [h[:2] for h in a.newton_hist] == [h[:2] for h in b.newton_hist]
    if mode != "binary":
        assert a.spec_hist[1] + a.chain_hist[1] > 0   # discards went through the hand-back too
    for name in ("P_est", "P_sch", "G", "diff"):
        assert torch_equal(getattr(a, name), getattr(b, name)), name
    a.step(write_sc=True); b.step(write_sc=True)   # and the run goes on identically
    for x, y in zip(a.result(), b.result()):
        np.testing.assert_array_equal(x, y)


def torch_equal(x, y):
    import torch
    return torch.equal(x, y)


@pytest.mark.parametrize("n,M,T", [(1, 1, 24), (5, 2, 24), (9, 9, 7), (40, 3, 33)])
def test_tiny_runs_match_oracle(gpu_lib, n, M, T):
    """Degenerate sizes through the whole engine (one residence, fewer residences than a
    wavefront holds, one residence per node, a slot count that fills no lane group): 20
    iterations, so that the native steady-state step with its folded home pass runs too."""
    from helpers import f32, oracle_homes
    from oracle import revs_oracle as ro
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(n, T, n_nodes=M, seed=n + T, binary_feasible=False, stress=0.9)
    w.load, w.cost = f32(w.load), f32(w.cost)
    e = _engine(w, "relaxed_exact")
    d = e.run(20)
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oracle_homes(w), w.Rn, w.node_of, w.cost, w.kappa, 20,
                                               w.vset, w.vlow, w.vhigh, mode="relaxed", util_eps=1e-10)
    P, S, C = e.result()
    assert e.spec_hist[0] > 0                                     # the native step did run
    assert np.abs(d - d_ref).max() < 1e-3 * max(1.0, d_ref.max())
    print(f"tiny run {n} x {T}: |S - oracle| {np.abs(S - S_ref).max():.2e} kW")
    assert np.abs(S - S_ref).max() < 5e-5 and np.abs(P - P_ref).max() < 5e-5


@pytest.mark.parametrize("mode,stress,T,nodes", [("pdhg", 1.02, 24, 200), ("relaxed_exact", 1.0, 24, 200),
                                                 ("binary", 0.5, 24, 200), ("pdhg", 1.02, 96, 200),
                                                 ("pdhg", 1.02, 24, 4096), ("relaxed_exact", 1.02, 24, 8192)])
def test_streaming_steady_state_equals_dense_product_path(gpu_lib, mode, stress, T, nodes):
    """The steady state as ONE launch per iteration -- rows judged inside the sweep's launch by
    the tree form of R p, later launches silenced on the device after a failed verdict, the
    host only keeping the queue full (revs_plan_stream_run) -- against the path that multiplies
    by the dense R on the matrix cores and reads every verdict on the host: same decisions
    (kept / discarded sweeps at the same iterations), same state bit for bit, in ragged
    chunks, through discards."""
    from helpers import f32
    from revs_admm_amd.engine import OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload((8000 if T == 24 else 3000) if nodes == 200 else 3 * nodes, T, n_nodes=nodes, seed=3,
                      binary_feasible=(mode == "binary"), stress=stress)
    w.load, w.cost = f32(w.load), f32(w.cost)
    a = _engine(w, mode, op=OperatorOptions(voltage="dense"))
    b = _engine(w, mode, op=OperatorOptions(voltage="tree", stream_burst=3, stream_burst_max=24))
    assert a._tree is None and b._tree is not None
    # (feeders of more than REVS_TREE_SWEEP_MAX nodes: the verdict launches AND -- round 4 -- the Newton
    # evaluations' row launches use the bigger workgroup shapes; only the chained iteration's fused launches
    # are 256 x 8 positions)
    assert b._tree_newton == (nodes <= 2048) and b._tree_eval and b._block == 32
    for chunk in (1, 7, 30, 2, 50):
        a.run_steps(chunk)
        b.run_steps(chunk)
        assert a.iteration == b.iteration
    assert a.spec_hist == b.spec_hist and a.spec_hist[0] > 40, (a.spec_hist, b.spec_hist)
    if mode != "binary":
        assert a.spec_hist[1] > 0                       # discards were crossed
    assert a.op_iters_hist == b.op_iters_hist
    # (same decisions at the same iterations; the values agree to the last bits of a float: the tree
    # engine also judges its Newton evaluations' rows by the tree form, the dense engine by the
    # matrix-core product -- 2e-15 relative apart in v)
    for name in ("P_est", "P_sch", "G", "diff"):
        x, y = getattr(a, name).cpu().numpy(), getattr(b, name).cpu().numpy()
        np.testing.assert_allclose(x, y, rtol=0, atol=2e-5, err_msg=name)
    a.step(write_sc=True); b.step(write_sc=True)
    for x, y in zip(a.result(), b.result()):
        np.testing.assert_allclose(x, y, rtol=0, atol=2e-5)


_RAGGED = (1, 7, 30, 2, 50, 64, 11)


@pytest.mark.parametrize("mode,n,nodes,seed,stress,T,block,chunks", [
    ("pdhg", 20000, 512, 0, 1.0, 24, 40, (40, 400, 400)),          # rows start to bind 192 sweeps into a burst
    ("pdhg", 20000, 512, 0, 1.0, 24, 7, (40, 400, 400)),
    ("pdhg", 20000, 512, 0, 1.0, 24, 32, (40, 400, 400)),         # ... at the first iteration of a block
    ("pdhg", 20000, 512, 0, 1.0, 24, 192, (40, 400, 400)),
    ("relaxed_exact", 20000, 512, 0, 1.01, 24, 5, (40, 400, 400)),
    ("pdhg", 8000, 200, 0, 1.1, 24, 16, (40, 400, 400)),
    ("pdhg", 8000, 200, 3, 1.02, 24, 5, _RAGGED),                   # a failed verdict at the head of a call
    ("binary", 8000, 200, 3, 0.5, 24, 7, _RAGGED),
    ("pdhg", 3000, 200, 3, 1.02, 96, 4, _RAGGED)])
@pytest.mark.parametrize("overlap,inner", [(False, 4), (True, 16), (True, 1), (False, 3), (True, 32)])
def test_block_verdicts_equal_per_launch_verdicts(gpu_lib, mode, n, nodes, seed, stress, T, block, chunks, overlap, inner):
    """The block form of the streaming loop (the default, sharded or not) -- `block` iterations run
    unjudged, their node sums go to a ring, one launch judges the whole block, `inner` consecutive
    iterations are ONE launch that keeps the residences' state in registers, and a failed iteration
    inside a block is undone from the set of buffers the block started from
    (revs_plan_stream_run_blocks) -- against the loop where every
    launch judges itself: same kept / discarded iterations, same memory bit for bit (profiles,
    carried PDHG multipliers, node sums handed to the next call), through failures at the head
    of a call and deep inside a burst (sweeps behind the failed one had run and are undone).
    `overlap`: the verdicts of a block on a second stream beside the sweeps of the next block."""
    from helpers import f32
    from revs_admm_amd.engine import OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(n, T, n_nodes=nodes, seed=seed, binary_feasible=(mode == "binary"), stress=stress)
    small = chunks is _RAGGED
    if small:
        w.load, w.cost = f32(w.load), f32(w.cost)
    kw = dict(stream_burst=16, stream_burst_max=64) if small else {}
    a = _engine(w, mode, op=OperatorOptions(stream_block_single=False, **kw))
    b = _engine(w, mode, op=OperatorOptions(stream_block=block, stream_block_single=True,
                                            stream_overlap=overlap, stream_inner=inner, **kw))
    assert a._block == 0 and b._block == block
    for chunk in chunks:
        a.run_steps(chunk)
        b.run_steps(chunk)
        assert a.iteration == b.iteration
        assert a.stream_calls == b.stream_calls
        for name in ("P_est", "P_sch", "G", "diff") + (("pdhg_dual",) if mode == "pdhg" else ()):
            assert torch_equal(getattr(a, name), getattr(b, name)), (name, a.iteration)
        if a._fused_ready:
            assert b._fused_ready and torch_equal(a._fused_p, b._fused_p)
            assert torch_equal(a.P_est_new, b.P_est_new)          # the prepared estimate P_est[k+1]
    assert a.spec_hist == b.spec_hist and a.chain_hist == b.chain_hist and a.spec_hist[0] > 60
    assert a.op_iters_hist == b.op_iters_hist and a.newton_hist == b.newton_hist
    failed = [(c, k) for c, k in b.stream_calls if k < c]
    if mode != "binary":
        assert failed, b.stream_calls
    if block in (192, 32) and mode == "pdhg" and n == 20000:    # the first iteration of the next block
        assert (400, 192) in failed, failed
    elif not small:     # ... inside a block: the sweeps behind the failed one had run
        assert any(k > block and k % block != 0 for c, k in failed), failed
    a.step(write_sc=True); b.step(write_sc=True)
    for x, y in zip(a.result(), b.result()):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("mode,block", [("pdhg", 0), ("relaxed_exact", 0), ("binary", 0), ("pdhg", 5),
                                        ("relaxed_exact", 32), ("binary", 8)])
def test_run_collects_every_iterations_diff_on_the_device(gpu_lib, mode, block):
    """AdmmEngine.run (what lpsolver.solve_ADMM calls) lets the steady-state launches write each
    iteration's diff into a history on the device and fetches it in one piece; the rows must be
    what a loop of step() reads back iteration by iteration -- through the transient, kept and
    discarded sweeps, and the last iteration's S and C."""
    from helpers import f32
    from revs_admm_amd.engine import OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(8000, 24, n_nodes=200, seed=3, binary_feasible=(mode == "binary"),
                      stress=0.5 if mode == "binary" else 1.02)
    w.load, w.cost = f32(w.load), f32(w.cost)
    K = 70
    a = _engine(w, mode)
    ref = np.zeros((K, a.n), np.float32)
    for k in range(K):
        a.step(write_sc=(k == K - 1))
        ref[k] = a.diff.cpu().numpy()[a.inv_perm]
    b = _engine(w, mode, op=OperatorOptions(stream_block=max(block, 1), stream_block_single=block > 0))
    d = b.run(K)
    assert b.spec_hist == a.spec_hist and b.spec_hist[0] > 30      # the streaming loop did the work
    assert len(b.stream_calls) < 15                                # ... in bursts, not iteration by iteration
    if block:
        # the convergence record: max_h diff[h] of every streamed iteration, folded on the device by
        # the sweeps and the verdict launches, equals the maximum of that iteration's row
        assert 30 < len(b.max_diff) <= b.spec_hist[0]
        for k, v in b.max_diff.items():
            assert v == float(d[k - 1].max()), (k, v, float(d[k - 1].max()))
    if mode == "pdhg":
        # (a discarded PDHG sweep leaves other warm-start multipliers when it ran inside a burst)
        np.testing.assert_allclose(d, ref, rtol=0, atol=2e-6)
    else:
        np.testing.assert_array_equal(d, ref)
    for x, y in zip(a.result(), b.result()):
        np.testing.assert_allclose(x, y, rtol=0, atol=0 if mode != "pdhg" else 2e-5)


def test_status_flags_surface_through_run_steps(gpu_lib):
    """A PDHG residence that stops at its iteration cap, or a residence whose window cannot
    reach 90 % SOC, is reported by the sweeps themselves (status bits OR-ed into a word the
    host reads) and raised at the next synchronisation point -- also when the iterations ran
    inside run_steps."""
    from revs_admm_amd import _lib
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(3000, 24, n_nodes=100, seed=5, binary_feasible=False, stress=0.9)
    e = _engine(w, "pdhg", pdhg={"max_iter": 8})
    e.run_steps(12)
    with pytest.raises(_lib.RevsError, match="REVS_ENOTCONV"):
        e.result()
    e = _engine(w, "pdhg")
    e.run_steps(12)
    e.result()                                          # the default cap is never reached here
    st = e.status.cpu().numpy()
    assert ((st & 3) == 0).all() and (st >> 8).max() < 4000
    w.homes["end"][7:9] = w.homes["start"][7:9] + 1     # one slot cannot deliver the energy
    w.homes["ev"][7:9] = 1
    e = _engine(w, "pdhg")
    e.run_steps(3)
    with pytest.raises(_lib.RevsError, match="No solution found"):
        e.residuals()


@pytest.mark.parametrize("case", ["binding", "stress6", "binary", "golden"])
def test_native_newton_solve_equals_python_loop(gpu_lib, golden, feeder_R, case):
    """revs_plan_newton_solve -- the operator's Newton solve (lpsolver.py:163-238 through its dual) as one
    native call: evaluations, models, Armijo line search, stopping and hand-off tests in the library --
    against the Python loop it mirrors (OperatorOptions(native_newton=False)): the same iterates, the same
    Newton bookkeeping, bit for bit; on synthetic feeders (few rows: the small model; stress 6: a third of
    the (residence, slot) pairs clamped, up to ten Newton iterations per solve), with on/off chargers, and
    on the reference's 121144 feeder (up to 69 binding rows per slot: the general model)."""
    from conftest import golden_homes
    from helpers import f32
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions, pack_homes
    from revs_admm_amd.synthetic import make_workload
    runs = []
    for native in (True, False):
        op = OperatorOptions(native_newton=native)
        if case == "golden":
            z, fd = golden
            oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
            n = oh.LOAD.shape[0]
            e = AdmmEngine(f32(z["tariff_shift6"]), pack_homes(oh.ev, 4.8, 20.0, 0.2, 11, 23), f32(oh.LOAD), np.arange(n),
                           feeder_R, kappa=5.0, vset=1.03, vlow=0.95, vhigh=1.05, mode="binary", op=op)
            iters = 15
        else:
            mode, stress, nh, nn = {"binding": ("pdhg", 1.3, 6000, 150), "stress6": ("relaxed_exact", 6.0, 8000, 200),
                                    "binary": ("binary", 1.0, 6000, 150)}[case]
            w = make_workload(nh, 24, n_nodes=nn, seed=5, binary_feasible=(mode == "binary"), stress=stress)
            w.load, w.cost = f32(w.load), f32(w.cost)
            e = _engine(w, mode, op=op)
            iters = 30
        d = e.run(iters)
        runs.append((d, e.result(), e.yd[0].cpu().numpy(), list(e.newton_hist), list(e.op_iters_hist), list(e.model_calls),
                     list(e.chain_hist), list(e.spec_hist)))
    a, b = runs
    assert max(h[0] for h in a[3]) >= (3 if case in ("stress6", "golden") else 1), a[3]      # real Newton solves on the way
    if case == "golden":
        assert a[5][1] > 0                                  # ... through the general model (more than 8 candidates)
    assert a[3] == b[3] and a[4] == b[4] and a[5] == b[5] and a[6] == b[6] and a[7] == b[7]
    np.testing.assert_array_equal(a[0], b[0])
    for x, y in zip(a[1], b[1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a[2], b[2])


@pytest.mark.gpu
def test_status_or_and_the_report_after_the_first_iteration(gpu_lib):
    """revs_status_or: the OR of bits 0-2 of the status words in a word of pinned memory (what check_status reads
    instead of fetching the array).  AdmmEngine.run reports a residence without a solution after the first iteration
    as the reference does (lpsolver.py:153-155) -- the reduction is enqueued behind that iteration's sweep and read
    behind the next one, whichever path that one takes."""
    import ctypes as C
    import torch
    from revs_admm_amd import _lib
    from revs_admm_amd._lib import check, ptr
    from revs_admm_amd.synthetic import make_workload
    dev = torch.device("cuda:0")
    flag = torch.zeros(2, dtype=torch.int32, pin_memory=True)
    dp = C.c_void_p()
    check(gpu_lib.revs_host_device_ptr(flag.data_ptr(), C.byref(dp)), "revs_host_device_ptr")
    for n, bits in ((1, 0), (1000, 0), (100_001, 4), (100_001, 1 | 2), (5, 2), (300_000, (3 << 8) | 4)):
        st = torch.full((n,), 7 << 8, dtype=torch.int32, device=dev)       # (PDHG pass counts in the high bits: ignored)
        if bits:
            st[n // 2] |= bits & 7
            st[n - 1] |= bits
        flag.zero_()
        check(gpu_lib.revs_status_or(n, ptr(st), dp.value, torch.cuda.current_stream(dev).cuda_stream), "revs_status_or")
        torch.cuda.synchronize()
        assert int(flag[0]) == (bits & 7) and int(flag[1]) == 0
    w = make_workload(3000, 24, n_nodes=100, seed=5, binary_feasible=False, stress=0.9)
    e = _engine(w, "pdhg")
    assert e.run(6, history=False) == 6                                 # (a feasible run: nothing raised)
    w.homes["end"][7:9] = w.homes["start"][7:9] + 1                     # one slot cannot deliver the energy
    w.homes["ev"][7:9] = 1
    for it in (1, 2, 12):
        e = _engine(w, "pdhg")
        with pytest.raises(_lib.RevsError, match="No solution found"):
            e.run(it, history=False)
        # ... after the second iteration at the latest (iter_max = 1: at the end of the run)
        assert e.iteration <= 2


@pytest.mark.gpu
def test_preallocation_changes_nothing(gpu_lib):
    """OperatorOptions(preallocate=False) -- pools, ring, events and the chained iteration's buffers on first use, as
    before round 5 -- walks the same trajectory bit for bit, through transient, chained iterations and streaming."""
    from revs_admm_amd.engine import OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(6000, 24, n_nodes=256, seed=2, binary_feasible=False, stress=1.2)
    runs = []
    for pre in (True, False):
        e = _engine(w, "pdhg", op=OperatorOptions(preallocate=pre))
        e.run_steps(45)
        runs.append((e.get_state(), list(e.op_iters_hist), e.spec_hist[0] + e.chain_hist[0]))
    for a, b in zip(runs[0][0], runs[1][0]):
        assert np.array_equal(a, b)
    assert runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2] and runs[0][2] > 0
