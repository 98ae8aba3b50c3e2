"""The sharded native loop with REAL partial sums (lpsolver.py:254-287: the loop whose iterates all
ranks must share).  Two processes, both on cuda:0, each with a node-aligned half of the residences,
run the real kernels and the real revs_plan_stream_run / revs_plan_stream_run_blocks; the node sums travel
through the library's hook communicator (revs_comm_create_hook) over gloo.  State after every chunk
must equal the one-process run bit for bit -- and must NOT when the collective is dropped or
applied to the wrong extent (negative controls)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
_RAGGED = (1, 7, 30, 2, 50, 64, 11)
_SMALL_BURST = dict(stream_burst=16, stream_burst_max=64)

CASES = [
    # rows start to bind ~190 sweeps into a burst: sweeps behind the failed one have run
    dict(name="deep", mode="pdhg", n=20000, nodes=512, seed=0, stress=1.0, T=24, chunks=(40, 400, 400),
         configs=[dict(tag="b32ov", block=32, overlap=True, hook="gloo"),
                  dict(tag="b32", block=32, overlap=False, hook="gloo"),
                  dict(tag="b4ov", block=4, overlap=True, hook="gloo"),
                  dict(tag="b1", block=1, overlap=False, hook="gloo"),
                  dict(tag="noop", block=32, overlap=True, hook="noop"),
                  dict(tag="slice", block=32, overlap=False, hook="first_slice")]),
    # a failed verdict at the head of a call, ragged chunks
    dict(name="head", mode="pdhg", n=8000, nodes=200, seed=3, stress=1.02, T=24, chunks=_RAGGED, f32=True,
         burst=_SMALL_BURST,
         configs=[dict(tag="b4", block=4, overlap=False, hook="gloo"),
                  dict(tag="b32ov", block=32, overlap=True, hook="gloo"),
                  dict(tag="b1", block=1, overlap=False, hook="gloo")]),
    dict(name="binary", mode="binary", n=8000, nodes=200, seed=3, stress=0.5, T=24, chunks=_RAGGED, f32=True,
         burst=_SMALL_BURST,
         configs=[dict(tag="b4ov", block=4, overlap=True, hook="gloo"),
                  dict(tag="b32", block=32, overlap=False, hook="gloo")]),
    # rows that keep binding: the folded chain (revs_plan_chain_fold_run) with residences sharded -- ONE all-reduce
    # of both folded sum arrays (8 M T doubles: {p, N, q, 0} per node and slot, twice) per iteration; PDHG residences and the reference's on/off chargers
    dict(name="fold", mode="pdhg", n=8000, nodes=200, seed=3, stress=1.3, T=24, chunks=(60, 1, 47, 40), f32=True, fold=True,
         configs=[dict(tag="b32ov", block=32, overlap=True, hook="gloo"),
                  dict(tag="noop", block=32, overlap=True, hook="noop")]),
    dict(name="fold_binary", mode="binary", n=8000, nodes=200, seed=3, stress=1.0, T=24, chunks=(50, 40, 30), f32=True,
         fold=True,
         configs=[dict(tag="b32", block=32, overlap=False, hook="gloo"),
                  dict(tag="noop", block=32, overlap=False, hook="noop")]),
    dict(name="t96", mode="pdhg", n=3000, nodes=200, seed=3, stress=1.02, T=96, chunks=_RAGGED, f32=True,
         burst=_SMALL_BURST,
         configs=[dict(tag="b4ov", block=4, overlap=True, hook="gloo"),
                  dict(tag="b1", block=1, overlap=False, hook="gloo")]),
]


@pytest.fixture(scope="module")
def two_rank_runs(gpu_lib, tmp_path_factory):
    """Both ranks run every case and configuration once (one pair of processes for the module)."""
    out = tmp_path_factory.mktemp("sharded")
    port = 29700 + os.getpid() % 2000
    procs = []
    for rank in range(2):
        spec = dict(rank=rank, world=2, port=port, outdir=str(out), cases=CASES)
        path = out / f"spec{rank}.json"
        path.write_text(json.dumps(spec))
        env = dict(os.environ, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2", MKL_NUM_THREADS="2")
        log = open(out / f"rank{rank}.log", "w")
        procs.append((subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(path)],
                                       stdout=log, stderr=subprocess.STDOUT, env=env), log))
    rcs = []
    for p, log in procs:
        try:
            rcs.append(p.wait(timeout=900))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
        log.close()
    logs = "\n".join((out / f"rank{r}.log").read_text()[-3000:] for r in range(2))
    assert rcs == [0, 0], logs
    return out


def _reference(case):
    """The one-process run, every launch judging itself (no blocks, no communicator; the binding
    steady state through the folded chain, as the sharded engine runs it since round 4)."""
    sys.path.insert(0, HERE)
    from sharded_worker import make_case, run_chunks
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    w = make_case(case)
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset, vlow=w.vlow,
                   vhigh=w.vhigh, mode=case["mode"], feeder=w.feeder,
                   op=OperatorOptions(stream_block_single=False, **(case.get("burst") or {})))
    assert e._block == 0 and e._comm is None
    return w, run_chunks(e, case["chunks"], case["mode"])


def _mismatch(case, ref, shards):
    """Names of the quantities in which the two shards differ from the one-process run."""
    bad = []
    names = ("P_est", "P_sch", "G", "diff") + (("pdhg_dual",) if case["mode"] == "pdhg" else ())
    for r in shards:
        lo, hi = int(r["lo"]), int(r["hi"])
        if "error" in r.files:
            bad.append("P_sch: " + str(r["error"]))
            continue
        for i in range(len(case["chunks"])):
            if int(r[f"iter_{i}"]) != int(ref[f"iter_{i}"]):
                bad.append(f"iter_{i}")
            for name in names:
                if not np.array_equal(r[f"{name}_{i}"], ref[f"{name}_{i}"][lo:hi]):
                    bad.append(f"{name}_{i}")
        for k in ("stream_calls", "spec_hist", "chain_hist", "op_iters"):
            if not np.array_equal(r[k], ref[k]):
                bad.append(k)
    return bad


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_two_ranks_with_real_partial_sums_equal_one_rank(two_rank_runs, case):
    w, ref = _reference(case)
    failed = [(int(c), int(k)) for c, k in ref["stream_calls"] if k < c]
    mt = w.M * w.T
    if case.get("fold"):
        assert ref["chain_hist"][0] > 30, ref["chain_hist"]    # the binding steady state did the work
    else:
        if case["mode"] != "binary":
            assert failed, ref["stream_calls"]             # failed verdicts were crossed
        assert ref["spec_hist"][0] > 60
    for cfg in case["configs"]:
        shards = [np.load(two_rank_runs / f"{case['name']}_{cfg['tag']}_r{r}.npz") for r in range(2)]
        assert int(shards[0]["lo"]) == 0 and int(shards[0]["hi"]) == int(shards[1]["lo"]) > 0
        assert int(shards[1]["hi"]) == case["n"] > int(shards[1]["lo"])
        bad = _mismatch(case, ref, shards)
        if cfg["hook"] == "gloo":
            assert not bad, (cfg, bad[:8])
            # both ranks issued the same collectives; with blocks, whole blocks in one call
            assert np.array_equal(shards[0]["hook_calls"], shards[1]["hook_calls"])
            sizes = shards[0]["hook_calls"]
            slice_ = mt + 2 * 64          # a ring slice: the node sums, then REVS_DMAX_SLOTS partial maxima of diff per rank
            if case.get("fold"):
                # the folded chain ran sharded: one collective of both folded arrays per chained iteration
                assert (sizes == 8 * mt).sum() >= ref["chain_hist"][0] - 20, ((sizes == 8 * mt).sum(), ref["chain_hist"])
            elif cfg["block"] > 1:
                blocks = sizes[(sizes >= slice_) & (sizes % slice_ == 0)]
                assert len(blocks) > 3 and blocks.max() <= cfg["block"] * slice_
                assert blocks.max() == min(cfg["block"], max(case["chunks"])) * slice_ or case["chunks"] is _RAGGED
                assert (sizes[sizes >= mt] % mt == 0).sum() + len(blocks) == (sizes >= mt).sum()
            else:
                assert (sizes[sizes >= mt] == mt).sum() > 60
        else:
            # negative controls: without the sums of the other rank (or with only one slice of
            # them) the verdicts are taken on wrong voltages -- the run must come out different
            assert bad, cfg
            assert any(b.startswith(("P_sch", "G_", "stream_calls", "spec_hist", "chain_hist", "op_iters")) for b in bad), bad[:8]


# ---- world size 8: eight logical ranks (threads) in one process on the one GPU, the library's hook communicator with
# nranks = 8 (revs_admm_amd.comm.LocalRanks) ----
CASES8 = [
    dict(name="deep8", mode="pdhg", n=40000, nodes=512, seed=0, stress=1.0, T=24, chunks=(40, 400, 400), align=8,
         configs=[dict(tag="b32ov", block=32, overlap=True, hook="local"),
                  dict(tag="b4", block=4, overlap=False, hook="local")]),
    dict(name="head8", mode="pdhg", n=16000, nodes=400, seed=3, stress=1.02, T=24, chunks=_RAGGED, f32=True, align=8,
         burst=_SMALL_BURST,
         configs=[dict(tag="b32ov", block=32, overlap=True, hook="local"),
                  dict(tag="b1", block=1, overlap=False, hook="local")]),
    dict(name="t96x8", mode="pdhg", n=8000, nodes=400, seed=3, stress=1.02, T=96, chunks=_RAGGED, f32=True, align=4,
         burst=_SMALL_BURST,
         configs=[dict(tag="b32ov", block=32, overlap=True, hook="local")]),
]


@pytest.fixture(scope="module")
def eight_rank_runs(gpu_lib, tmp_path_factory):
    out = tmp_path_factory.mktemp("sharded8")
    spec = dict(local=True, world=8, outdir=str(out), cases=CASES8)
    path = out / "spec.json"
    path.write_text(json.dumps(spec))
    env = dict(os.environ, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2", MKL_NUM_THREADS="2")
    with open(out / "worker.log", "w") as log:
        p = subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(path)], stdout=log,
                             stderr=subprocess.STDOUT, env=env)
        try:
            rc = p.wait(timeout=1200)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = -9
    assert rc == 0, (out / "worker.log").read_text()[-4000:]
    return out


@pytest.mark.parametrize("case", CASES8, ids=[c["name"] for c in CASES8])
def test_eight_logical_ranks_equal_one_rank(eight_rank_runs, case):
    """World size 8 on one GPU: every rank's residences after every chunk equal the one-process run bit for bit, all
    ranks issued the same collectives, and a block collective carries 64 x 8 partial maxima of diff per ring slice."""
    w, ref = _reference(case)
    mt = w.M * w.T
    failed = [(int(c), int(k)) for c, k in ref["stream_calls"] if k < c]
    assert failed, ref["stream_calls"]                     # failed verdicts were crossed
    for cfg in case["configs"]:
        shards = [np.load(eight_rank_runs / f"{case['name']}_{cfg['tag']}_r{r}.npz") for r in range(8)]
        edges = [int(s["lo"]) for s in shards] + [int(shards[-1]["hi"])]
        assert edges[0] == 0 and edges[-1] == case["n"] and all(a < b for a, b in zip(edges[:-1], edges[1:]))
        assert [int(s["hi"]) for s in shards[:-1]] == edges[1:-1]
        bad = _mismatch(case, ref, shards)
        assert not bad, (cfg, bad[:8])
        for s in shards[1:]:
            assert np.array_equal(s["hook_calls"], shards[0]["hook_calls"])
        sizes = shards[0]["hook_calls"]
        slice_ = mt + 8 * 64              # a ring slice: the node sums, then REVS_DMAX_SLOTS partial maxima of diff per rank
        if cfg["block"] > 1:
            blocks = sizes[(sizes >= slice_) & (sizes % slice_ == 0)]
            assert len(blocks) > 3 and blocks.max() <= cfg["block"] * slice_
        else:
            assert (sizes[sizes >= mt] == mt).sum() > 60


@pytest.mark.parametrize("transport", ["rccl", "share-gpu"])
def test_bench_two_ranks_through_its_own_launcher(gpu_lib, transport):
    """`python bench.py --gpus 2` end to end: the self-launcher (Popen children, never exec), sharding, the native
    block loop with the library's communicator in it, max over ranks, rank 0's JSON line -- against the one-process
    run of the same total workload (--scaling strong): the same ADMM residuals after the same 95 iterations.
    "rccl": one rank per GPU over the library-owned RCCL communicator -- needs two GPUs, skipped on the one-GPU test
    box; "share-gpu": both ranks on cuda:0 over the hook communicator (gloo), the same code path with a host-staged
    transport."""
    import torch
    if transport == "rccl" and torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the round-end test box has one)")
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    common = [sys.executable, bench, "--steps", "20", "--bursts", "3", "--homes", "20000", "--nodes", "512", "--scaling", "strong",
              "--no-extras", "--no-cpu-baseline", "--no-converge", "--lanes", "0"]
    env = dict(os.environ, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2", MKL_NUM_THREADS="2")
    lines = []
    for extra in (["--gpus", "1"], ["--gpus", "2"] + (["--share-gpu"] if transport == "share-gpu" else [])):
        r = subprocess.run(common + extra, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines.append(json.loads([ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")][-1]))
    one, two = lines
    assert (one["n_gpus"], two["n_gpus"]) == (1, 2) and two["config"]["homes_total"] == one["config"]["homes_total"] == 20000
    assert two["config"]["homes_per_gpu"] == 10000 and two["config"]["launcher"] == "bench.py spawned the ranks itself"
    assert two["breakdown"]["steady_state_steps_kept_discarded"] == one["breakdown"]["steady_state_steps_kept_discarded"] == [60, 0]
    for k in ("admm_residual_primal", "admm_residual_dual", "admm_max_diff"):      # (PDHG residences group by wavefront: not bit for bit)
        assert two["breakdown"][k] == pytest.approx(one["breakdown"][k], rel=2e-3), k
    c = two["collective"]
    assert c is not None and c["collective_ms_per_block"] > 0 and c["iterations_in_that_block"] == 20 and c["bytes"] == 20 * (512 * 24 + 128) * 8
    assert one["collective"] is None
