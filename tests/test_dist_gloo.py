"""N > 1 path: residences sharded over ranks, node space replicated, ONE collective
per inner iteration (all-reduce of the node aggregate) -- run under gloo with
world_size 2 on CPU through the test double, and compared with the 1-rank run."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, mode, solver, stress, iters, out):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    from fake_kernels import FakeKernels
    from helpers import f32
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = make_workload(240, 12, n_nodes=24, seed=6, stress=stress, binary_feasible=(mode == "binary"))
    w.load, w.cost = f32(w.load), f32(w.cost)
    lo, hi = w.shard(rank, world)
    e = AdmmEngine(w.cost, w.homes[lo:hi], w.load[lo:hi], w.node_of[lo:hi], w.Rn, kappa=w.kappa,
                   vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode=mode, device="cpu",
                   group=dist.group.WORLD, _kernels=FakeKernels(),
                   op=OperatorOptions(solver=solver))
    # global node counts came from the all-reduce in the constructor
    assert (e.node_counts == np.bincount(w.node_of, minlength=w.M)).all()
    d = e.run(iters)
    P, S, C = e.result()
    rp, rd, dmax, conv = e.residuals(1e-4)
    v = e.voltage().numpy().copy()
    np.savez(out.format(rank=rank), d=d, S=S, P=P, lo=lo, hi=hi, res=[rp, rd, dmax], v=v,
             iters=e.op_iters_hist, spec=e.spec_hist, chain=e.chain_hist)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,solver,stress,iters", [
    ("relaxed_exact", "newton", 1.3, 3), ("binary", "newton", 1.3, 3),
    ("relaxed_exact", "admm", 1.3, 3), ("binary", "admm", 1.3, 3),
    # long enough for the steady state: speculative home sweeps kept AND discarded, with only
    # p exchanged in the speculative evaluation
    ("relaxed_exact", "newton", 1.02, 12),
    # ... and for the binding steady state: Newton iterations enqueued whole, kept and redone
    ("relaxed_exact", "newton", 1.3, 30)])
def test_two_ranks_equal_one_rank(tmp_path, mode, solver, stress, iters):
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    from fake_kernels import FakeKernels
    from helpers import f32
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    from revs_admm_amd.synthetic import make_workload
    port = 29500 + (os.getpid() % 2000)
    out = str(tmp_path / "r{rank}.npz")
    mp.spawn(_worker, args=(2, port, mode, solver, stress, iters, out), nprocs=2, join=True)
    w = make_workload(240, 12, n_nodes=24, seed=6, stress=stress, binary_feasible=(mode == "binary"))
    w.load, w.cost = f32(w.load), f32(w.cost)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                   vlow=w.vlow, vhigh=w.vhigh, mode=mode, device="cpu", _kernels=FakeKernels(),
                   op=OperatorOptions(solver=solver))
    d1 = e.run(iters)
    P1, S1, C1 = e.result()
    rp, rd, dmax, _ = e.residuals(1e-4)
    r = [np.load(out.format(rank=k)) for k in range(2)]
    assert (r[0]["lo"], r[0]["hi"], r[1]["lo"], r[1]["hi"]) == (0, 120, 120, 240)
    d2 = np.concatenate([r[0]["d"], r[1]["d"]], axis=1)
    S2 = np.concatenate([r[0]["S"], r[1]["S"]], axis=0)
    # same algorithm, same data; only the order of the floating-point node sums differs
    assert np.abs(d2 - d1).max() < 1e-5 and np.abs(S2 - S1).max() < 1e-4
    assert list(r[0]["iters"]) == list(r[1]["iters"])          # ranks stop together
    assert list(r[0]["spec"]) == list(r[1]["spec"]) == list(e.spec_hist)
    assert list(r[0]["chain"]) == list(r[1]["chain"]) == list(e.chain_hist)
    if iters == 12:
        assert e.spec_hist[0] > 0 and e.spec_hist[1] > 0
    if iters == 30:
        assert min(e.chain_hist) > 0
    np.testing.assert_allclose(r[0]["res"], [rp, rd, dmax], rtol=1e-4)
    np.testing.assert_allclose(r[0]["res"], r[1]["res"], rtol=0, atol=0)
    # voltage profile R.(aggregate load): identical on both ranks after the all-reduce
    np.testing.assert_array_equal(r[0]["v"], r[1]["v"])
    np.testing.assert_allclose(r[0]["v"], e.voltage().numpy(), rtol=1e-5, atol=1e-7)


def _eps_worker(rank, world, port, eps, out):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    from fake_kernels import FakeKernels
    from helpers import f32
    from revs_admm_amd.engine import AdmmEngine
    from revs_admm_amd.synthetic import make_workload
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = make_workload(240, 12, n_nodes=24, seed=6, stress=1.02)
    w.load, w.cost = f32(w.load), f32(w.cost)
    lo, hi = w.shard(rank, world)
    e = AdmmEngine(w.cost, w.homes[lo:hi], w.load[lo:hi], w.node_of[lo:hi], w.Rn, kappa=w.kappa,
                   vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode="relaxed_exact", device="cpu",
                   group=dist.group.WORLD, _kernels=FakeKernels())
    d = e.run(80, eps=eps, patience=3)
    np.savez(out.format(rank=rank), d=d, at=-1 if e.converged_at is None else e.converged_at,
             md=[e.max_diff[k] for k in sorted(e.max_diff)], own=d.max(axis=1))
    dist.barrier()
    dist.destroy_process_group()


def test_run_eps_stops_every_rank_at_the_same_iteration(tmp_path):
    """AdmmEngine.run(eps=) with residences sharded: iterations that go through step() record
    max_h diff over EVERY rank's residences (a rank deciding on its own residences'
    maximum stops alone and the others hang in their next all-reduce)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    from fake_kernels import FakeKernels
    from helpers import f32
    from revs_admm_amd.engine import AdmmEngine
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(240, 12, n_nodes=24, seed=6, stress=1.02)
    w.load, w.cost = f32(w.load), f32(w.cost)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow,
                   vhigh=w.vhigh, mode="relaxed_exact", device="cpu", _kernels=FakeKernels())
    d_all = e.run(40)
    # an eps that one half of the residences reaches well before the other half does
    half = [d_all[:, :120].max(axis=1), d_all[:, 120:].max(axis=1)]
    gap = [k for k in range(8, 40) if min(half[0][k], half[1][k]) < 0.7 * max(half[0][k], half[1][k])]
    assert gap, "workload: the halves' maxima never differ enough to tell the two rules apart"
    k = gap[len(gap) // 2]
    eps = float(np.sqrt(half[0][k] * half[1][k]))
    port = 29500 + ((os.getpid() + 977) % 2000)
    out = str(tmp_path / "e{rank}.npz")
    mp.spawn(_eps_worker, args=(2, port, eps, out), nprocs=2, join=True)
    r = [np.load(out.format(rank=q)) for q in range(2)]
    assert r[0]["d"].shape[0] == r[1]["d"].shape[0]              # same number of iterations
    assert int(r[0]["at"]) == int(r[1]["at"]) and int(r[0]["at"]) > 0
    np.testing.assert_array_equal(r[0]["md"], r[1]["md"])       # the records are global
    both = np.maximum(r[0]["own"], r[1]["own"])
    np.testing.assert_allclose(r[0]["md"], both[:len(r[0]["md"])], rtol=1e-6)
    # ... and the one-rank run stops at the same iteration
    e1 = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow,
                    vhigh=w.vhigh, mode="relaxed_exact", device="cpu", _kernels=FakeKernels())
    d1 = e1.run(80, eps=eps, patience=3)
    assert e1.converged_at == int(r[0]["at"]) and d1.shape[0] == r[0]["d"].shape[0]


def test_local_ranks_allreduce_and_mismatch():
    """revs_admm_amd.comm.LocalRanks (logical ranks as threads of one process: the transport of the eight-rank GPU test):
    sum / max / min over 8 ranks, the same bits on every rank; a rank that reduces another size than its peers ends in an
    error on every rank, not in a hang."""
    import threading
    from revs_admm_amd.comm import LocalRanks
    W = LocalRanks(8, timeout=20.0)
    res, errs = [None] * 8, [None] * 8

    def work(r):
        g = W.rank(r)
        a = np.arange(5, dtype=np.float64) * (r + 1) + 0.1 * r
        b, c = a.copy(), a.copy()
        g.allreduce_host(a, 0); g.allreduce_host(b, 2); g.allreduce_host(c, 3)
        res[r] = (a, b, c)
    ts = [threading.Thread(target=work, args=(r,)) for r in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    base = [np.arange(5, dtype=np.float64) * (r + 1) + 0.1 * r for r in range(8)]
    want = base[0].copy()
    for x in base[1:]:
        want = want + x
    for r in range(8):
        np.testing.assert_array_equal(res[r][0], want)
        np.testing.assert_array_equal(res[r][1], np.max(base, axis=0))
        np.testing.assert_array_equal(res[r][2], np.min(base, axis=0))
    assert W.calls[0] == [5, 5, 5] and all(c == W.calls[0] for c in W.calls)
    W2 = LocalRanks(3, timeout=5.0)

    def bad(r):
        try:
            W2.rank(r).allreduce_host(np.zeros(4 if r else 7), 0)
        except Exception as ex:              # RuntimeError on the ranks that compared sizes, BrokenBarrierError on the others
            errs[r] = ex
    ts = [threading.Thread(target=bad, args=(r,)) for r in range(3)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert all(errs[r] is not None for r in range(3))


@pytest.mark.parametrize("mode,stress,iters", [("relaxed_exact", 1.02, 12), ("binary", 1.3, 4)])
def test_four_local_ranks_equal_one_rank(mode, stress, iters):
    """The engine over comm.LocalRanks (four logical ranks, a thread each, the numpy test double of the C ABI): same
    trajectory as one rank, every rank takes the same decisions (the sharded driver logic without gloo or processes --
    what tests/test_gpu_sharded.py::test_eight_logical_ranks_equal_one_rank runs on the GPU with the real kernels)."""
    import threading
    sys.path.insert(0, HERE)
    from fake_kernels import FakeKernels
    from helpers import f32
    from revs_admm_amd.comm import LocalRanks
    from revs_admm_amd.engine import AdmmEngine
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(240, 12, n_nodes=24, seed=6, stress=stress, binary_feasible=(mode == "binary"))
    w.load, w.cost = f32(w.load), f32(w.cost)
    world = 4
    W = LocalRanks(world, timeout=60.0)
    out, errs = [None] * world, [None] * world

    def work(r):
        try:
            lo, hi = w.shard(r, world)
            e = AdmmEngine(w.cost, w.homes[lo:hi], w.load[lo:hi], w.node_of[lo:hi], w.Rn, kappa=w.kappa, vset=w.vset,
                           vlow=w.vlow, vhigh=w.vhigh, mode=mode, device="cpu", group=W.rank(r), _kernels=FakeKernels())
            assert (e.node_counts == np.bincount(w.node_of, minlength=w.M)).all()
            d = e.run(iters)
            out[r] = (d, e.result()[1], list(e.op_iters_hist), list(e.spec_hist), list(e.chain_hist))
        except Exception:
            import traceback
            errs[r] = traceback.format_exc()
            W.abort()
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not any(errs), [x for x in errs if x]
    e1 = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow, vhigh=w.vhigh,
                    mode=mode, device="cpu", _kernels=FakeKernels())
    d1 = e1.run(iters)
    S1 = e1.result()[1]
    d4 = np.concatenate([o[0] for o in out], axis=1)
    S4 = np.concatenate([o[1] for o in out], axis=0)
    assert np.abs(d4 - d1).max() < 1e-5 and np.abs(S4 - S1).max() < 1e-4
    for o in out[1:]:
        assert o[2:] == out[0][2:]
    assert out[0][2] == list(e1.op_iters_hist) and out[0][3] == list(e1.spec_hist)
