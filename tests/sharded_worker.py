"""One rank of the two-process tests of the sharded streaming loop (tests/test_gpu_sharded.py).

    python tests/sharded_worker.py spec.json

Both ranks use cuda:0 (RCCL refuses two ranks on one device, so the node sums travel through the
library's hook communicator -- revs_comm_create_hook -- over a gloo group); each owns a
node-aligned half of the residences and runs the REAL kernels and the REAL native loops
(revs_plan_stream_run / revs_plan_stream_run_blocks).  After every chunk of run_steps the rank stores its
shard's state, in the caller's home order, for the parent to compare with the one-process run.
Hooks: "gloo" = the real all-reduce; "noop" = nothing is exchanged; "first_slice" = the all-reduce
is done, but only the first M x T slice of a multi-slice buffer keeps its sums (the others are left
at zero: a collective applied to the wrong extent).  The last two are negative controls."""
import datetime
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def node_aligned_split(node_of, world, align=32):
    """Shard boundaries at node boundaries (every node's residences then sit on ONE rank: its sums
    are formed in the one-process order and the all-reduce adds exact zeros) that are also a
    multiple of 32 residences (a PDHG residence iterates until the slowest of the residences sharing
    its wavefront has converged, so the grouping into wavefronts must be the one-process one) --
    the sharded run must then equal the one-process run bit for bit.  Nearest such boundary to
    equal shares."""
    import numpy as np
    node_of = np.asarray(node_of)
    n = len(node_of)
    ok = np.flatnonzero((node_of[1:] != node_of[:-1])) + 1
    ok = ok[ok % align == 0]          # (align: 32 = a workgroup at T = 24; 8 = a wavefront there, which is what the grouping needs)
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(ok[np.argmin(np.abs(ok - (n * r) // world))]))
    cuts.append(n)
    assert (np.diff(cuts) > 0).all(), cuts
    return np.asarray(cuts)


def make_case(case):
    from helpers import f32
    from revs_admm_amd.synthetic import make_workload
    w = make_workload(case["n"], case["T"], n_nodes=case["nodes"], seed=case["seed"],
                      binary_feasible=(case["mode"] == "binary"), stress=case["stress"])
    if case.get("f32"):
        w.load, w.cost = f32(w.load), f32(w.cost)
    return w


def run_chunks(e, chunks, mode):
    """State after every chunk, caller's home order."""
    import numpy as np
    out = {}
    for i, c in enumerate(chunks):
        e.run_steps(c)
        for name in ("P_est", "P_sch", "G", "diff") + (("pdhg_dual",) if mode == "pdhg" else ()):
            out[f"{name}_{i}"] = getattr(e, name).cpu().numpy()[e.inv_perm]
        out[f"iter_{i}"] = np.asarray(e.iteration)
    out["stream_calls"] = np.asarray(e.stream_calls, np.int64).reshape(-1, 2)
    out["spec_hist"] = np.asarray(e.spec_hist)
    out["chain_hist"] = np.asarray(e.chain_hist)
    out["op_iters"] = np.asarray(e.op_iters_hist)
    return out


def main_local(spec):
    """`world` LOGICAL ranks in this one process, a thread each (revs_admm_amd.comm.LocalRanks): the library's hook
    communicator with nranks = world -- the 64 x ranks partial-maximum slots of a ring slice, the block collectives, the
    3 : 1 split of a burst's tail -- beyond the two processes the other cases use (a GPU box admits few processes on
    its card; eight ranks of the real thing are the driver's 8-GPU run)."""
    import threading
    import traceback
    import numpy as np
    import torch
    from revs_admm_amd.comm import LocalRanks
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    world = spec["world"]
    torch.cuda.set_device(0)
    for case in spec["cases"]:
        w = make_case(case)
        cuts = node_aligned_split(w.node_of, world, align=case.get("align", 32))
        counts = np.bincount(w.node_of, minlength=w.M)
        for cfg in case["configs"]:
            ranks = LocalRanks(world, timeout=180.0)
            outs, errs = [None] * world, [None] * world

            def work(r, _ranks=ranks, _outs=outs, _errs=errs, _cfg=cfg):
                try:
                    torch.cuda.set_device(0)
                    lo, hi = int(cuts[r]), int(cuts[r + 1])
                    kw = dict(case.get("burst") or {})
                    e = AdmmEngine(w.cost, w.homes[lo:hi], w.load[lo:hi], w.node_of[lo:hi], w.Rn, kappa=w.kappa,
                                   vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode=case["mode"], device="cuda:0",
                                   group=_ranks.rank(r), node_counts=counts, feeder=w.feeder,
                                   op=OperatorOptions(stream_block=_cfg["block"], stream_overlap=_cfg["overlap"], **kw))
                    assert e._comm is not None and e._plan is not None and e._tree is not None
                    assert e._block == (_cfg["block"] if _cfg["block"] > 1 else 0)
                    out = run_chunks(e, case["chunks"], case["mode"])
                    out["lo"], out["hi"] = np.asarray(lo), np.asarray(hi)
                    _outs[r] = out
                    del e
                except Exception:
                    _errs[r] = traceback.format_exc()
                    _ranks.abort()

            threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            bad = [f"rank {r}:\n{x}" for r, x in enumerate(errs) if x]
            if bad:
                print("\n".join(bad), flush=True)
                raise SystemExit(1)
            for r in range(world):
                outs[r]["hook_calls"] = np.asarray(ranks.calls[r], np.int64)
                np.savez(os.path.join(spec["outdir"], f"{case['name']}_{cfg['tag']}_r{r}.npz"), **outs[r])
            torch.cuda.empty_cache()


def main():
    spec = json.load(open(sys.argv[1]))
    if spec.get("local"):
        return main_local(spec)
    rank, world = spec["rank"], spec["world"]
    import numpy as np
    import torch
    import torch.distributed as dist
    from revs_admm_amd.engine import AdmmEngine, OperatorOptions
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(spec["port"]))
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    torch.cuda.set_device(0)
    for case in spec["cases"]:
        w = make_case(case)
        cuts = node_aligned_split(w.node_of, world)
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        counts = np.bincount(w.node_of, minlength=w.M)
        mt = w.M * w.T
        for cfg in case["configs"]:
            calls = []

            def real(a, op, _calls=calls):
                _calls.append(len(a))
                dist.all_reduce(torch.from_numpy(a), op={0: dist.ReduceOp.SUM, 2: dist.ReduceOp.MAX,
                                                         3: dist.ReduceOp.MIN}[op])

            def noop(a, op, _calls=calls):
                _calls.append(len(a))

            def first_slice(a, op, _calls=calls):
                real(a, op)
                if len(a) > mt and len(a) % mt == 0:
                    a[mt:] = 0.0

            hook = {"gloo": real, "noop": noop, "first_slice": first_slice}[cfg["hook"]]
            kw = dict(case.get("burst") or {})
            e = AdmmEngine(w.cost, w.homes[lo:hi], w.load[lo:hi], w.node_of[lo:hi], w.Rn, kappa=w.kappa,
                           vset=w.vset, vlow=w.vlow, vhigh=w.vhigh, mode=case["mode"], device="cuda:0",
                           group=dist.group.WORLD, node_counts=counts, feeder=w.feeder, comm_hook=hook,
                           op=OperatorOptions(stream_block=cfg["block"], stream_overlap=cfg["overlap"], **kw))
            assert e._comm is not None and e._plan is not None and e._tree is not None
            assert e._block == (cfg["block"] if cfg["block"] > 1 else 0)
            try:
                out = run_chunks(e, case["chunks"], case["mode"])
            except Exception as ex:      # a negative control may also end in an error: that is a mismatch too
                if cfg["hook"] == "gloo":
                    raise
                out = {"error": np.asarray(str(ex))}
            out["lo"], out["hi"] = np.asarray(lo), np.asarray(hi)
            out["hook_calls"] = np.asarray(calls, np.int64)
            np.savez(os.path.join(spec["outdir"], f"{case['name']}_{cfg['tag']}_r{rank}.npz"), **out)
            del e
            torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
