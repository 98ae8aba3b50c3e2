"""Host-side mirror of the reference interface: records, R matrix, readers/writers,
and the engine's driver logic on the test double (tests/fake_kernels.py).  CPU only."""
import numpy as np
import os
import pytest

from oracle import revs_oracle as ro


def _graph(fd, z):
    import networkx as nx
    g = nx.Graph()
    for i, (nid, lab) in enumerate(zip(z["node_id"], fd.label)):
        g.add_node(int(nid), label=lab.decode())
    for u, v, r in zip(fd.edge_u, fd.edge_v, fd.edge_r):
        g.add_edge(int(z["node_id"][u]), int(z["node_id"][v]), r=float(r))
    return g


def test_pack_homes_slot_counts():
    """nmin/nmax of revs_home_t == the oracle's reading of lpsolver.py:101-109,
    including the reference's own parameters (0.2, 20 kWh, 4.8 / 3.6 kW -> 3 / 4)."""
    from revs_admm_amd.engine import pack_homes
    rec = pack_homes([True, True, False, True], [4.8, 3.6, 0, 7.2], [20, 20, 1, 40],
                     [0.2, 0.2, 0, 0.5], 11, 23)
    assert rec["nmin"].tolist() == [3, 4, 0, 3] and rec["nmax"].tolist() == [3, 4, 0, 2]
    oh = ro.homes_from_records(np.zeros((4, 24)), rec)
    nmin, nmax = ro.slot_count_bounds(oh)
    assert nmin.tolist() == rec["nmin"].tolist() and nmax.tolist() == rec["nmax"].tolist()
    rng = np.random.default_rng(0)
    n = 5000
    rec = pack_homes(rng.random(n) < 0.7, rng.choice([3.6, 4.8, 7.2, 11.0], n),
                     rng.choice([20., 40., 60., 75.], n), rng.uniform(0, 0.95, n), 10, 24)
    nmin, nmax = ro.slot_count_bounds(ro.homes_from_records(np.zeros((n, 24)), rec))
    assert (nmin == rec["nmin"]).all() and (nmax == rec["nmax"]).all()


def test_compute_Rmat_matches_reference_formula(golden):
    """Product compute_Rmat(graph) == literal 2 F D F^T (lpsolver.py:17-26)."""
    from revs_admm_amd.lpsolver import compute_Rmat
    z, fd = golden
    R = compute_Rmat(_graph(fd, z))
    assert np.abs(R - ro.compute_Rmat(fd)).max() < 1e-15


def test_combine_result_roundtrip(golden):
    """combine_result writes the reference's out/ text format: re-parse and compare
    with the stored distributed file's numbers."""
    import importlib.util, os
    from revs_admm_amd.extract import combine_result
    z, fd = golden
    tag = "dis_a90_r4800"
    res, ev = z["res_id"].tolist(), z[tag + "_ev_homes"].tolist()
    P_res = {h: z[tag + "_P_res"][i].tolist() for i, h in enumerate(res)}
    P_ev = {h: z[tag + "_P_ev"][i].tolist() for i, h in enumerate(ev)}
    SOC = {h: z[tag + "_SOC"][i].tolist() for i, h in enumerate(ev)}
    diff = {k + 1: {h: float(z[tag + "_diff"][i, k]) for i, h in enumerate(ev)} for k in range(15)}
    txt = combine_result(P_res, P_ev, SOC, ev, diff)
    spec = importlib.util.spec_from_file_location(
        "mk", os.path.join(os.path.dirname(__file__), "golden", "make_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "revs_combine.txt")
    open(p, "w").write(txt)
    s = mk.read_result(p)
    assert list(s["Residence Usage Profile"]) == res
    np.testing.assert_array_equal(np.array([s["EV Convergence over Iterations"][h] for h in ev]),
                                  z[tag + "_diff"])
    np.testing.assert_array_equal(np.array([s["EV Charger Usage Profile"][h] for h in ev]),
                                  z[tag + "_P_ev"])


def test_get_homes_ev_param_and_arrays(golden):
    from revs_admm_amd.extract import get_homes_ev_param
    from revs_admm_amd.lpsolver import homes_to_arrays
    z, fd = golden
    g = _graph(fd, z)
    res = z["res_id"].tolist()
    loads = {h: z["LOAD"][i].tolist() for i, h in enumerate(res)}
    ev = z["dis_a90_r4800_ev_homes"]
    homes = get_homes_ev_param(loads, g, ev, 4.8, 20, 0.2, 11, 23)
    assert list(homes) == res and sum(1 for h in res if homes[h]["EV"]) == 267
    assert homes[int(ev[0])]["EV"] == dict(rating=4.8, capacity=20.0, initial=0.2, start=11, end=23)
    load, rec = homes_to_arrays(homes, res)
    assert load.shape == (1126, 24) and rec["ev"].sum() == 267
    assert (rec["nmin"][rec["ev"] == 1] == 3).all()


def test_ev_charge_curve():
    from revs_admm_amd.ev_charge import charge
    assert charge(0.0) == 0.0 and abs(charge(3.5) - 189.0 * (1 - np.exp(-6.9077))) < 1e-12
    assert charge(0.0, P0=50) == 50.0


def test_synthetic_workload_invariants():
    from revs_admm_amd.synthetic import make_workload, radial_R
    w = make_workload(3000, 24, seed=4)
    h = w.homes
    ev = h["ev"] == 1
    assert (h["nmin"][ev] <= h["nmax"][ev]).all() and (h["nmin"][ev] <= (h["end"] - h["start"])[ev]).all()
    assert (np.diff(w.node_of) >= 0).all() and w.Rn.shape == (w.M, w.M)
    assert np.abs(w.Rn - w.Rn.T).max() == 0 and np.linalg.eigvalsh(w.Rn).min() > 0
    # radial_R == oracle's tree form on the same feeder
    fd = ro.Feeder(np.array([b"S"] + [b"T"] * w.M), np.where(w.parent < 0, 0, w.parent + 1),
                   np.arange(1, w.M + 1), w.edge_r)
    assert np.abs(ro.compute_Rmat_tree(fd) - w.Rn).max() < 1e-12 * w.Rn.max()


@pytest.mark.parametrize("solver", ["newton", "admm"])
@pytest.mark.parametrize("mode,omode", [("binary", "binary"), ("relaxed_exact", "relaxed")])
def test_engine_driver_on_fake_kernels(mode, omode, solver):
    """AdmmEngine's host logic (operator driver -- dual Newton with its line search, or the
    ADMM forms with stopping rule and rho adaptation -- and the P_est swap) with the numpy
    test double == oracle solve_ADMM."""
    from fake_kernels import FakeKernels
    from helpers import f32, oracle_homes
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(150, 12, n_nodes=20, seed=2, stress=1.3, binary_feasible=(mode == "binary"))
    w.load, w.cost = f32(w.load), f32(w.cost)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                   vlow=w.vlow, vhigh=w.vhigh, mode=mode, device="cpu", _kernels=FakeKernels(),
                   op=OperatorOptions(solver=solver))
    d = e.run(4)
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oracle_homes(w), w.Rn, w.node_of, w.cost, w.kappa, 4,
                                               w.vset, w.vlow, w.vhigh, mode=omode)
    P, S, Cs = e.result()
    if solver == "newton":       # the rows bind: Newton iterations were needed, none failed
        assert set(e.op_path_hist) == {"dual"} and max(n for n, _, _ in e.newton_hist) >= 1
    else:
        assert max(e.op_iters_hist) > 25
    if mode == "binary":
        same = np.abs(S - S_ref).max(1) == 0
        assert same.mean() > 0.97
    else:
        assert np.abs(S - S_ref).max() < 1e-3 and np.abs(d - d_ref).max() < 1e-3


@pytest.mark.parametrize("stress", [0.9, 2.0, 4.0])
def test_dual_newton_on_fake_kernels(stress):
    """Driver logic of the dual Newton path (candidate sets, model solves, Armijo steps,
    warm start of the multipliers) from a feeder whose rows barely bind to one stressed so
    hard that whole subtrees are clamped at zero -- same answers as the oracle's
    home-space ADMM, to the oracle's own tolerance."""
    from fake_kernels import FakeKernels
    from helpers import f32, oracle_homes
    from revs_admm_amd.engine import AdmmEngine
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(150, 12, n_nodes=20, seed=2, stress=stress, binary_feasible=False)
    w.load, w.cost = f32(w.load), f32(w.cost)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                   vlow=w.vlow, vhigh=w.vhigh, mode="relaxed_exact", device="cpu",
                   _kernels=FakeKernels())
    d = e.run(6)
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oracle_homes(w), w.Rn, w.node_of, w.cost, w.kappa, 6,
                                               w.vset, w.vlow, w.vhigh, mode="relaxed")
    assert set(e.op_path_hist) == {"dual"}
    assert max(n for n, _, _ in e.newton_hist) >= 1                       # the rows do bind
    if stress >= 2.0:
        assert e.P_est.min().item() == 0.0                               # clamped residences
    assert np.abs(e.result()[1] - S_ref).max() < 1e-5 and np.abs(d - d_ref).max() < 1e-6


def test_newton_handoff_to_admm_on_fake_kernels():
    """A Newton solve that cannot finish (here: not allowed more than one iteration) hands the
    ADMM iteration to the ADMM forms and starts from zero multipliers next time; the run still
    follows the oracle."""
    from fake_kernels import FakeKernels
    from helpers import f32, oracle_homes
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(150, 12, n_nodes=20, seed=2, stress=2.0, binary_feasible=False)
    w.load, w.cost = f32(w.load), f32(w.cost)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                   vlow=w.vlow, vhigh=w.vhigh, mode="relaxed_exact", device="cpu",
                   _kernels=FakeKernels(), op=OperatorOptions(newton_max=1))
    d = e.run(5)
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oracle_homes(w), w.Rn, w.node_of, w.cost, w.kappa, 5,
                                               w.vset, w.vlow, w.vhigh, mode="relaxed")
    assert {"node", "home"} & set(e.op_path_hist) and max(h[0] for h in e.newton_hist) == 1
    assert np.abs(e.result()[1] - S_ref).max() < 1e-5 and np.abs(d - d_ref).max() < 1e-6


def test_speculative_sweep_on_fake_kernels():
    """step() launches the home sweep behind the operator's first evaluation when the last
    solve needed no Newton iteration.  Kept or discarded, the trajectory is the one of the
    non-speculative driver, bit for bit."""
    from fake_kernels import FakeKernels
    from helpers import f32
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(150, 12, n_nodes=20, seed=2, stress=1.02, binary_feasible=False)
    w.load, w.cost = f32(w.load), f32(w.cost)
    runs = []
    for spec in (True, False):
        e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                       vlow=w.vlow, vhigh=w.vhigh, mode="relaxed_exact", device="cpu",
                       _kernels=FakeKernels(), op=OperatorOptions(speculate=spec))
        runs.append((e.run(12), e.result(), e))
    (d1, r1, e1), (d0, r0, e0) = runs
    assert e1.spec_hist[0] > 0 and e1.spec_hist[1] > 0 and e0.spec_hist == [0, 0]
    np.testing.assert_array_equal(d1, d0)
    for a, b in zip(r1, r0):
        np.testing.assert_array_equal(a, b)
    assert [h[0] for h in e1.newton_hist] == [h[0] for h in e0.newton_hist]


@pytest.mark.parametrize("mode,stress", [("binary", 1.1), ("relaxed_exact", 1.3)])
def test_chained_newton_iteration_on_fake_kernels(mode, stress):
    """Binding steady state: after a solve of exactly one Newton iteration on the small model,
    step() enqueues evaluation, model, step, evaluation and the sweep before reading anything
    (_chain_launch).  Kept or redone, the trajectory is the unchained driver's, bit for bit."""
    from fake_kernels import FakeKernels
    from helpers import f32
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(150, 12, n_nodes=20, seed=2, stress=stress, binary_feasible=(mode == "binary"))
    w.load, w.cost = f32(w.load), f32(w.cost)
    runs = []
    for chain in (True, False):
        e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                       vlow=w.vlow, vhigh=w.vhigh, mode=mode, device="cpu",
                       _kernels=FakeKernels(), op=OperatorOptions(chain=chain))
        runs.append((e.run(30), e.result(), e))
    (d1, r1, e1), (d0, r0, e0) = runs
    assert min(e1.chain_hist) > 0 and e0.chain_hist == [0, 0], (e1.chain_hist, e1.newton_hist)
    np.testing.assert_array_equal(d1, d0)
    for a, b in zip(r1, r0):
        np.testing.assert_array_equal(a, b)
    assert [h[0] for h in e1.newton_hist] == [h[0] for h in e0.newton_hist]


def test_run_steps_without_a_plan_is_a_loop_of_steps():
    """Without the native plan (no GPU here) run_steps(k) is k calls of step()."""
    from fake_kernels import FakeKernels
    from helpers import f32
    from revs_admm_amd.engine import AdmmEngine
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(120, 12, n_nodes=15, seed=4, stress=1.02, binary_feasible=False)
    w.load, w.cost = f32(w.load), f32(w.cost)
    mk = lambda: AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                            vlow=w.vlow, vhigh=w.vhigh, mode="relaxed_exact", device="cpu",
                            _kernels=FakeKernels())
    a, b = mk(), mk()
    for _ in range(9):
        a.step(write_sc=False)
    b.run_steps(4)
    b.run_steps(5)
    assert a.iteration == b.iteration == 9 and a.op_iters_hist == b.op_iters_hist
    np.testing.assert_array_equal(a.P_sch.numpy(), b.P_sch.numpy())
    np.testing.assert_array_equal(a.G.numpy(), b.G.numpy())


@pytest.mark.parametrize("stress,paths", [(0.9, {"node"}), (2.0, {"node", "home"})])
def test_operator_paths_on_fake_kernels(stress, paths):
    """Driver logic of the two ADMM operator paths (OperatorOptions.solver = "admm"): the node-space fast path is kept while no
    residence is pushed to zero, and the general home-space ADMM takes over (for the rest of
    the run) on a feeder stressed so hard that some are -- same answers as the oracle."""
    from fake_kernels import FakeKernels
    from helpers import f32, oracle_homes
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(150, 12, n_nodes=20, seed=2, stress=stress, binary_feasible=False)
    w.load, w.cost = f32(w.load), f32(w.cost)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                   vlow=w.vlow, vhigh=w.vhigh, mode="relaxed_exact", device="cpu",
                   _kernels=FakeKernels(), op=OperatorOptions(solver="admm"))
    d = e.run(6)
    d_ref, P_ref, S_ref, C_ref = ro.solve_ADMM(oracle_homes(w), w.Rn, w.node_of, w.cost, w.kappa, 6,
                                               w.vset, w.vlow, w.vhigh, mode="relaxed")
    assert set(e.op_path_hist) == paths and max(e.op_iters_hist) > 0     # the rows do bind
    if "home" in paths:           # after a wasted fast solve the retry backs off (2, 4, ... steps)
        k = e.op_path_hist.index("home")
        assert set(e.op_path_hist[k:]) == {"home"} and e.P_est.min().item() == 0.0
    assert np.abs(e.result()[1] - S_ref).max() < 1e-5 and np.abs(d - d_ref).max() < 1e-6


def test_operator_not_converged_is_an_error_not_an_answer():
    """An operator QP that stops at max_iter above its tolerance raises REVS_ENOTCONV from
    step() / run() instead of handing a half-projected estimate to the residences."""
    from fake_kernels import FakeKernels
    from helpers import f32
    from revs_admm_amd import _lib
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(150, 12, n_nodes=20, seed=2, stress=1.6, binary_feasible=False)
    w.load, w.cost = f32(w.load), f32(w.cost)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                   vlow=w.vlow, vhigh=w.vhigh, mode="relaxed_exact", device="cpu",
                   _kernels=FakeKernels(),
                   op=OperatorOptions(solver="admm", max_iter=25, eps=1e-13, calibrate=False))
    with pytest.raises(_lib.RevsError, match="REVS_ENOTCONV"):
        e.run(6)
    assert 1 <= e.iteration < 6                     # (the first iterations need no projection)


def test_feeder_as_a_tree_reproduces_R(golden, feeder_R):
    """feeder_tree / tree_voltage_host (the host side of revs_tree_t, and the numpy restatement of
    the kernel's three prefix sums): R p from the tree == the dense product, on the 121144 feeder
    (1691 tree nodes, 1126 constraint rows, extracted from a networkx graph as solve_ADMM does),
    on a synthetic forest with nodes that carry no residence, and through the padding to a
    multiple of 8 positions; a tree that does not match Rn is refused by the engine."""
    import networkx as nx
    from revs_admm_amd.engine import feeder_tree, tree_voltage_host
    from revs_admm_amd.lpsolver import feeder_arrays
    from revs_admm_amd.synthetic import make_workload
    z, fd = golden
    g = nx.Graph()
    for nid, lab in zip(z["node_id"], fd.label):
        g.add_node(int(nid), label=lab.decode())
    for u, v, r in zip(fd.edge_u, fd.edge_v, fd.edge_r):
        g.add_edge(int(z["node_id"][u]), int(z["node_id"][v]), r=float(r))
    res = [n for n in g if g.nodes[n]["label"] == "H"]
    par, er, cons = feeder_arrays(g, res)
    assert len(par) == 1691 and (cons >= 0).sum() == 1126 and (par < 0).sum() >= 1
    tr = feeder_tree(par, er, cons, np.ones(len(res), bool))
    assert tr["n"] == 1696 and tr["n"] % 8 == 0                    # padded
    rng = np.random.default_rng(0)
    p = rng.uniform(0, 3, (len(res), 5))
    ref = feeder_R @ p
    assert np.abs(tree_voltage_host(tr, p) - ref).max() < 1e-12 * np.abs(ref).max()
    # subtree ranges are nested or disjoint, ends sorted, cle consistent
    j = np.arange(tr["n"])
    assert (tr["end"] > j).all() and (np.diff(tr["end"][tr["eo"]]) >= 0).all()
    assert (tr["cle"] == np.searchsorted(np.sort(tr["end"]), j, side="right")).all()
    w = make_workload(900, 24, n_nodes=300, seed=4)
    chk = rng.random(300) < 0.6
    tr = feeder_tree(*w.feeder, chk)
    p = rng.uniform(0, 3, (300, 3)) * chk[:, None]
    ref = (w.Rn @ p) * chk[:, None]
    got = tree_voltage_host(tr, p)
    assert np.abs(got - ref).max() < 1e-12 * np.abs(ref).max() and (got[~chk] == 0).all()
    assert (tr["src"] >= 0).sum() == chk.sum() and (w.parent < 0).sum() > 1     # a forest
    with pytest.raises(ValueError, match="forest"):
        feeder_tree(np.array([1, 0]), np.ones(2), np.arange(2), np.ones(2, bool))   # a cycle
    # the engine checks a feeder against Rn before it trusts it
    from fake_kernels import FakeKernels
    from revs_admm_amd.engine import AdmmEngine
    kw = dict(kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode="relaxed_exact",
              device="cpu", _kernels=FakeKernels())
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, feeder=w.feeder, **kw)
    assert e._tree is not None and e._plan is None
    with pytest.raises(ValueError, match="does not reproduce Rn"):
        AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, feeder=(w.parent, w.edge_r * 1.01, np.arange(300)), **kw)


def test_feeder_pack_rejects_rows_beyond_16_bits():
    """revs_tree_t.pack holds four 16-bit fields per position: a constraint row beyond them must be
    refused on the host (it would silently corrupt the neighbouring field on the device)."""
    from revs_admm_amd.feeder import feeder_tree
    par = np.array([-1, 0, 1, 2], np.int64)
    cons = np.array([0, 70000, -1, 3])
    with pytest.raises(ValueError, match="16-bit"):
        feeder_tree(par, np.ones(4), cons, np.ones(70001, bool))
    assert feeder_tree(par, np.ones(4), np.array([0, 65000, -1, 3]), np.ones(65001, bool))["n"] == 8


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no launcher: the parent spawns N fresh ranks (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* on 127.0.0.1), relays rank 0's one JSON line on stdout and
    exits non-zero when a rank fails.  --dry-run stops after the rendezvous (no GPU here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run",
                        "--scaling", "strong"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"dry_run": True, "n_gpus": 2, "sum_ranks": 1.0, "sum_local_ranks": 1.0,
                   "ranks_seen": 2.0, "scaling": "strong", "launcher": "self"}
    import torch
    if not torch.cuda.is_available():      # without a GPU every rank refuses: the parent must fail too
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode != 0 and "needs a ROCm GPU" in r.stderr and not r.stdout.strip()


def test_plot_result_warns_and_returns():
    """test-optimizer.py:55-58 ends in fx.plot_result(...): out of scope, but it must not kill the caller."""
    from revs_admm_amd.revs_fixture import REVS
    fx = REVS.__new__(REVS)
    with pytest.warns(RuntimeWarning, match="outside"):
        assert fx.plot_result(None, None, anything=1) is None


def test_tree_recovered_from_the_matrix(golden, feeder_R):
    """revs_admm_amd.feeder.tree_from_R: callers of lpsolver.solve_ADMM's operator hold R restricted to the
    residence nodes (lpsolver.py:184-189), not the network.  A radial feeder is recovered from that matrix alone
    -- junctions that carry no residence become extra tree nodes -- and reproduces it: the synthetic feeder
    (every node a row) exactly as given, the 121144 feeder's 1126 residence rows with fewer nodes than the
    network has (1691), within the tree form's 2048; a matrix that is not a feeder's is rejected or fails the
    product check the engine makes."""
    from revs_admm_amd.feeder import feeder_tree, tree_from_R, tree_voltage_host
    from revs_admm_amd.synthetic import make_workload
    rng = np.random.default_rng(0)
    w = make_workload(3000, 24, n_nodes=300, seed=3)
    par, er, cons = tree_from_R(w.Rn)
    assert len(par) == w.M and sorted(cons) == list(range(w.M))
    for R in (w.Rn, feeder_R):
        M = R.shape[0]
        tr = tree_from_R(R)
        assert tr is not None and M <= len(tr[0]) <= 2048 and (np.sort(tr[2][tr[2] >= 0]) == np.arange(M)).all()
        ft = feeder_tree(*tr, np.ones(M, bool))
        probe = rng.uniform(0.5, 1.5, (M, 3))
        ref = R @ probe
        assert np.abs(tree_voltage_host(ft, probe) - ref).max() < 1e-11 * np.abs(ref).max()
    assert len(tree_from_R(feeder_R)[0]) < 1691
    A = rng.uniform(0, 1, (40, 40))
    bad = tree_from_R(A @ A.T)
    if bad is not None:                                      # (no tree metric: whatever comes out does not reproduce it)
        ft = feeder_tree(*bad, np.ones(40, bool))
        probe = rng.uniform(0.5, 1.5, (40, 2))
        assert np.abs(tree_voltage_host(ft, probe) - (A @ A.T) @ probe).max() > 1e-6
    assert tree_from_R(np.array([[1.0, 2.0], [0.5, 1.0]])) is None      # not symmetric
