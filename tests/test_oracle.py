"""The oracle checked against itself (independent formulations) and against the
reference's stored results.  CPU only."""
import numpy as np
import pytest

from oracle import revs_oracle as ro


def _wl(n, T, seed, **kw):
    from helpers import oracle_homes
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(n, T, seed=seed, n_nodes=12, **kw)
    return w, oracle_homes(w)


def _state(w, seed):
    rng = np.random.default_rng(seed)
    n, T = w.load.shape
    ps = w.load + rng.uniform(0, 3, (n, T))
    return ps * rng.uniform(0.7, 1.1, (n, T)), ps, rng.normal(0, 2.0, (n, T))


def test_R_dense_equals_tree(golden):
    """compute_Rmat restated literally (lpsolver.py:17-26) == path-sum form."""
    z, fd = golden
    # a 300-node sub-tree keeps the dense inverse fast; the full feeder is checked too
    R = ro.compute_Rmat(fd)
    Rt = ro.compute_Rmat_tree(fd)
    assert R.shape == (1691, 1691)
    assert np.abs(R - Rt).max() < 1e-15
    assert np.abs(R - R.T).max() < 1e-18 and R.min() >= 0


@pytest.mark.parametrize("T", [6, 10, 12])
def test_binary_selection_is_the_miqp_optimum(T):
    """home_solve_binary (selection) vs literal enumeration of every on/off pattern."""
    w, oh = _wl(60, T, seed=T)
    # shrink the windows to the toy horizon
    oh.start[:] = np.minimum(oh.start, T // 3)
    oh.end[:] = T
    nmin, nmax = ro.slot_count_bounds(oh)
    pe, ps, gm = _state(w, T)
    p, s, g, st = ro.home_solve_binary(w.cost, oh, pe, ps, gm, w.kappa)
    best = ro.home_solve_binary_bruteforce(w.cost, oh, pe, ps, gm, w.kappa)
    obj = ro.home_objective(w.cost, oh, p, pe, ps, gm, w.kappa)
    feas = st == 0
    assert feas.sum() > 20
    np.testing.assert_allclose(obj[feas], best[feas], rtol=1e-12, atol=1e-12)
    assert np.isinf(best[~feas]).all()
    # SOC rows hold literally
    assert (s[feas][:, -1][oh.ev[feas]] >= 0.9 - 1e-9).all() and (s <= 1 + 1e-9).all()


@pytest.mark.parametrize("T", [24, 96])
def test_relaxed_exact_vs_pdhg(T):
    """Closed-form relaxed optimum == a long float64 PDHG run on the full SOC rows."""
    w, oh = _wl(300, T, seed=T, binary_feasible=False)
    pe, ps, gm = _state(w, T)
    p, s, g, st = ro.home_solve_relaxed(w.cost, oh, pe, ps, gm, w.kappa)
    p2, *_ , nit = ro.home_solve_relaxed_pdhg(w.cost, oh, pe, ps, gm, w.kappa, iters=20000,
                                             tol=1e-12)
    assert (st == 0).all()
    assert np.abs(p - p2).max() < 1e-7
    # KKT of the closed form: energy window respected, box respected
    Elo, Ehi = ro.energy_bounds(oh)
    e = p.sum(1)
    assert (e >= Elo - 1e-9).all() and (e <= Ehi + 1e-9).all()
    assert (p >= 0).all() and (p <= oh.rating[:, None] + 1e-12).all() and (p[~oh.window()] == 0).all()


def test_utility_solver_kkt():
    """Operator QP: ADMM answer satisfies the KKT conditions of lpsolver.py:163-238 and
    equals a second, independent method (dual projected gradient run to convergence
    on a small well-conditioned case)."""
    w, oh = _wl(40, 6, seed=5)
    pe, ps, gm = _state(w, 1)
    g0 = ro.utility_g0(pe, ps, gm, w.kappa)
    vlo, vhi = ro.voltage_limits(w.vset, w.vlow, w.vhigh)
    g, info = ro.utility_solve(w.Rn, w.node_of, g0, w.kappa, vlo, vhi, return_info=True)
    prim, stat, comp = ro.utility_kkt(w.Rn, w.node_of, g, g0, w.kappa, vlo, vhi,
                                      info["yv"], info["yb"])
    assert prim < 1e-9 and stat < 1e-8 and comp < 1e-8
    assert np.abs(g - np.maximum(g0, 0)).max() > 1e-3        # the voltage rows do bind
    # independent check: accelerated dual projected gradient
    M = w.Rn.shape[0]
    A = np.zeros((M, w.N)); A[w.node_of, np.arange(w.N)] = 1
    RA = w.Rn @ A
    Lc = np.linalg.norm(RA, 2) ** 2 / w.kappa
    mu = np.zeros((M, g0.shape[1])); y = mu.copy(); tk = 1.0
    for _ in range(200000):
        gg = np.maximum(0, g0 - RA.T @ y / w.kappa)
        mu_n = np.maximum(0, y + (RA @ gg - vhi) / Lc)
        tn = (1 + np.sqrt(1 + 4 * tk * tk)) / 2
        y = mu_n + (tk - 1) / tn * (mu_n - mu)
        if np.vdot(y - mu_n, mu_n - mu) > 0:
            y, tn = mu_n.copy(), 1.0
        if np.abs(mu_n - mu).max() < 1e-13 * max(1.0, np.abs(mu_n).max()):
            mu = mu_n
            break
        mu, tk = mu_n, tn
    g2 = np.maximum(0, g0 - RA.T @ mu / w.kappa)
    assert np.abs(g - g2).max() < 1e-6


def test_golden_individual(golden):
    """Stored individual-mode results (3 cases) are optimal for the restated model:
    same objective for every residence, same slot count, inside the window."""
    from conftest import golden_homes
    z, fd = golden
    for tag, rate in [("ind_a90_r4800", 4.8), ("ind_a70_r4800", 4.8), ("ind_a90_r3600", 3.6)]:
        oh, evi = golden_homes(z, tag, rate)
        p, s, g = ro.solve_residence(z["tariff_shift6"], oh)
        pref = np.zeros_like(p)
        pref[evi] = z[tag + "_P_ev"]
        o1 = ro.residence_objective(z["tariff_shift6"], oh, p)
        o2 = ro.residence_objective(z["tariff_shift6"], oh, pref)
        assert np.abs(o1 - o2).max() < 1e-12
        assert ((p > 0).sum(1) == (pref > 1e-6).sum(1)).all()
        np.testing.assert_allclose(z[tag + "_SOC"][:, -1], s[evi][:, -1], atol=1e-9)
        np.testing.assert_allclose(z[tag + "_P_res"], pref + oh.LOAD, atol=1e-9)


def test_golden_distributed_first_iteration(golden):
    """diff[1] of the stored distributed run (267 EV homes) to 1e-12: pins the home
    objective, the SOC slot count, the dual update and the residual definition."""
    from conftest import golden_homes
    z, fd = golden
    oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
    zero = np.zeros_like(oh.LOAD)
    p, s, g, st = ro.home_solve_binary(z["tariff_shift6"], oh, zero, zero, zero, 5.0)
    # iteration 1: the operator's answer to an all-zero state is zero (lpsolver.py:244-259)
    diff1 = np.linalg.norm(0.0 - g, axis=1) / 24
    np.testing.assert_allclose(diff1[evi], z["dis_a90_r4800_diff"][:, 0], rtol=0, atol=1e-12)


def test_golden_distributed_final_is_feasible(golden):
    """The stored final schedules satisfy the restated home constraints: 3 full-rate
    slots inside 11..22, SOC 0.2 -> 0.92, P_res = LOAD + P_ev."""
    from conftest import golden_homes
    z, fd = golden
    oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
    P = z["dis_a90_r4800_P_ev"]
    assert np.isin(np.round(P, 9), [0.0, 4.8]).all()
    assert ((P > 1e-6).sum(1) == 3).all()      # (solver noise ~1e-15 in the file)
    assert (np.abs(P[:, :11]) < 1e-9).all() and (np.abs(P[:, 23:]) < 1e-9).all()
    soc = 0.2 + np.cumsum(P, 1) / 20.0
    np.testing.assert_allclose(z["dis_a90_r4800_SOC"][:, 1:], soc, atol=1e-9)
    np.testing.assert_allclose(z["dis_a90_r4800_P_res"][evi], P + oh.LOAD[evi], atol=1e-9)


def test_dual_operator_solver_equals_admm_solver():
    """utility_solve_dual (working-set dual, certified by utility_kkt) == utility_solve (home-space
    ADMM on the literal QP) on a state where rows bind and residences are clamped."""
    w, oh = _wl(200, 6, seed=9, stress=1.6)
    pe, ps, gm = _state(w, 4)
    g0 = ro.utility_g0(pe, ps, gm, w.kappa) - 1.0            # some g0 < 0: the lb = 0 rows matter
    vlo, vhi = ro.voltage_limits(w.vset, w.vlow, w.vhigh)
    g_admm = ro.utility_solve(w.Rn, w.node_of, g0, w.kappa, vlo, vhi, eps=1e-12, max_iter=200000)
    g_dual, yv, yb = ro.utility_solve_dual(w.Rn, w.node_of, g0, w.kappa, vlo, vhi)
    assert (yv != 0).sum() > 0 and ((g_dual == 0) & (g0 < 0)).sum() > 0
    assert np.abs(g_admm - g_dual).max() < 1e-7
    # the negative-control variant (no lb) is a different model: it must differ here
    g_nolb, *_ = ro.utility_solve_dual(w.Rn, w.node_of, g0, w.kappa, vlo, vhi, nonneg=False)
    assert np.abs(g_nolb - g_dual).max() > 1e-3


def test_golden_distributed_trajectory_statistics(golden, feeder_R):
    """15 iterations on the 121144 feeder against the stored trajectory (267 EV residences x
    15 iterations, final schedules) on tie-robust statistics, under both consistent tie rules
    (earlier / later slot) -- and two negative controls, each a plausible misreading of
    lpsolver.py, which MUST fail the same bounds under both rules:
      * homes solved from the new estimate P_est[k+1] instead of P_est[k] (lpsolver.py:273);
      * the operator QP without Gurobi's default variable lower bound 0 (lpsolver.py:179-180).
    Also: a tie rule that is redrawn every iteration is far outside the band."""
    import functools
    from conftest import golden_homes
    from helpers import GOLDEN_BOUNDS, golden_trajectory_stats
    z, fd = golden
    oh, evi = golden_homes(z, "dis_a90_r4800", 4.8)
    n = oh.N

    def run(variant, tie, iters=15):
        hs = functools.partial(ro.home_solve_binary, tie=tie)
        d, P, S, C = ro.solve_ADMM(oh, feeder_R, np.arange(n), z["tariff_shift6"], 5.0, iters, 1.03,
                                   0.95, 1.05, mode="binary", util_method="dual", variant=variant,
                                   home_solver=hs)
        return d, S

    for tie in (None, "last"):
        d, S = run(None, tie)
        good = golden_trajectory_stats(d[:, evi], S[evi], z)
        for k, bound in GOLDEN_BOUNDS.items():
            assert good[k] <= bound, (tie, k, good[k], bound)
        for variant in ("new_estimate", "no_lb"):
            d, S = run(variant, tie)
            bad = golden_trajectory_stats(d[:, evi], S[evi], z)
            failed = [k for k, bound in GOLDEN_BOUNDS.items() if bad[k] > bound]
            assert len(failed) >= 4, (variant, tie, bad)          # not a marginal miss
            assert bad["mean"] > 1.4 * good["mean"] and bad["w1"] > 1.25 * good["w1"]
    d, S = run(None, np.random.default_rng(1))
    ref = z["dis_a90_r4800_diff"].T
    assert np.abs(d[:, evi].mean(1) / ref.mean(1) - 1).max() > 1.0


def test_central_lp(golden):
    """solve_central_lp: (i) with slack voltage rows it is every residence's own cheapest-slot
    optimum; (ii) with binding rows its answer is feasible and costs more; (iii) it is what the
    distributed iteration converges to (total cost within 1 % after 600 oracle iterations, from
    above); (iv) the MILP form (binary chargers) is bounded below by the LP."""
    w, oh = _wl(120, 12, seed=3, binary_feasible=False, stress=0.3)
    p, g, per_home, tot = ro.solve_central_lp(w.cost, oh, w.Rn, w.node_of, w.vset, w.vlow, w.vhigh)
    Elo, _ = ro.energy_bounds(oh)
    # own optimum: fill the cheapest window slots up to E_lo
    own = np.zeros(oh.N)
    for i in np.where(oh.ev)[0]:
        need, c = Elo[i], 0.0
        for t in np.argsort(np.where(oh.window()[i], w.cost, np.inf), kind="stable"):
            take = min(need, oh.rating[i])
            c += take * w.cost[t]
            need -= take
            if need <= 0:
                break
        own[i] = c
    np.testing.assert_allclose(per_home, own + oh.LOAD @ w.cost, rtol=1e-9)
    w2, oh2 = _wl(120, 12, seed=3, binary_feasible=False, stress=1.2)
    p2, g2, per2, tot2 = ro.solve_central_lp(w2.cost, oh2, w2.Rn, w2.node_of, w2.vset, w2.vlow, w2.vhigh)
    vlo, vhi = ro.voltage_limits(w2.vset, w2.vlow, w2.vhigh)
    A = np.zeros((w2.Rn.shape[0], oh2.N)); A[w2.node_of, np.arange(oh2.N)] = 1
    v = w2.Rn @ A @ g2
    assert v.max() <= vhi * (1 + 1e-7) and v.max() >= vhi * (1 - 1e-6)       # rows bind
    assert (p2.sum(1) >= Elo - 1e-7).all() and (p2 >= -1e-9).all() and (p2[~oh2.window()] == 0).all()
    assert tot2 > tot * (1 + 1e-4)
    d, P, S, C = ro.solve_ADMM(oh2, w2.Rn, w2.node_of, w2.cost, w2.kappa, 600, w2.vset, w2.vlow,
                               w2.vhigh, mode="relaxed", util_method="dual")
    dev_total = ((P @ w2.cost).sum() - tot2) / tot2
    assert -1e-3 < dev_total < 0.01, dev_total


def test_central_ref_model_on_the_golden_feeder(golden, feeder_R):
    """solve_central_ref -- lpsolver.solve_central's own model (no s_T >= 0.9 row, -R g between the
    limits) -- returns what the reference stored for the 121144 feeder: every charger off,
    P_res = LOAD; and with a negative price it switches on exactly the slots that pay."""
    from conftest import golden_homes
    z, fd = golden
    h, _ = golden_homes(z, "cen_a90_r4800", 4.8)
    p, g, tot = ro.solve_central_ref(z["tariff_shift6"], h, feeder_R, 1.03, 0.90, 1.05)
    assert (p == 0).all()
    np.testing.assert_allclose(g, z["cen_a90_r4800_P_res"], atol=1e-9)
    t2 = np.array(z["tariff_shift6"], float)
    t2[13] = -0.05
    p2, g2, tot2 = ro.solve_central_ref(t2, h, feeder_R, 1.03, 0.90, 1.05)
    assert (p2[h.ev, 13] == h.rating[h.ev]).all() and p2.sum() == pytest.approx(h.rating[h.ev].sum())


def test_central_milp_small():
    """Binary chargers (the reference's home model) in the centralized problem: a MILP through
    HiGHS; bounded below by its LP relaxation, schedules on/off at full rating, rows respected."""
    w3, oh3 = _wl(16, 24, seed=5, stress=1.05)
    pb, gb, perb, totb = ro.solve_central_lp(w3.cost, oh3, w3.Rn, w3.node_of, w3.vset, w3.vlow,
                                             w3.vhigh, binary=True, time_limit=60)
    pl, gl, perl, totl = ro.solve_central_lp(w3.cost, oh3, w3.Rn, w3.node_of, w3.vset, w3.vlow, w3.vhigh)
    assert totb >= totl - 1e-9
    assert np.isin(np.round(pb / np.where(oh3.ev, oh3.rating, 1.0)[:, None], 6), [0.0, 1.0]).all()
    nmin, nmax = ro.slot_count_bounds(oh3)
    non = (pb > 1e-9).sum(1)
    assert (non >= nmin).all() and (non <= nmax).all()
    vlo, vhi = ro.voltage_limits(w3.vset, w3.vlow, w3.vhigh)
    A = np.zeros((w3.Rn.shape[0], oh3.N)); A[w3.node_of, np.arange(oh3.N)] = 1
    assert (w3.Rn @ A @ gb).max() <= vhi * (1 + 1e-7)
