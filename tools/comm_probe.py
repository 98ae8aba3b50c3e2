"""The sharded streaming steady state rehearsed on one GPU (RCCL group of one rank): time per
iteration of run_steps with the library's communicator in the loop, and the bare cost of its
all-reduce (host time per call, stream time per call).  python tools/comm_probe.py [homes]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd._lib import check, ptr                 # noqa: E402
from revs_admm_amd.engine import AdmmEngine, OperatorOptions   # noqa: E402
from revs_admm_amd.synthetic import make_workload         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.update(RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
w = make_workload(n, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)


def run(group, unset=False, block=32, single=False, overlap=True):
    e = AdmmEngine(w.cost, w.homes, w.load, w.node_of, w.Rn, kappa=w.kappa, vset=w.vset,
                   vlow=w.vlow, vhigh=w.vhigh, mode="pdhg", feeder=w.feeder, group=group,
                   op=OperatorOptions(stream_block=block, stream_block_single=single, stream_overlap=overlap))
    e.run_steps(40)
    if unset:                       # the engine's group code paths without the call in the loop
        check(e.lib.revs_plan_set_comm(e._plan, None), "set_comm")
    torch.cuda.synchronize()
    out, calls = [], []
    inner = e._stream_run

    def timed_call(count):
        t0 = time.perf_counter()
        n = inner(count)
        calls.append((n, round((time.perf_counter() - t0) * 1e6 / max(n, 1), 1)))
        return n
    e._stream_run = timed_call
    for _ in range(2):
        t0 = time.perf_counter()
        e.run_steps(250)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 250 * 1e6)
    print("   native calls (iterations, us per iteration):", calls)
    print((("group, no call in the loop" if unset else "group") if group is not None else "alone")
          + f", verdicts by blocks of {e._block}{', second stream' * overlap}" * (e._block > 0), "us per iteration, blocks of 250:",
          " ".join(f"{x:.2f}" for x in out), "kept/discarded", e.spec_hist, flush=True)
    return e


run(None)
run(None, single=True, overlap=False)
run(None, single=True)
run(None, block=8, single=True)
run(dist.group.WORLD, block=1)
run(dist.group.WORLD, overlap=False)
run(dist.group.WORLD, block=8)
e = run(dist.group.WORLD)
buf = torch.zeros(2048 * 24, dtype=torch.float64, device="cuda:0")
for count in (2048 * 24, 32 * 2048 * 24):
    big = torch.zeros(count, dtype=torch.float64, device="cuda:0")
    for reps in (1, 200):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for _ in range(reps):
            check(e.lib.revs_comm_allreduce_f64(e._comm, ptr(big), count, 0, e.stream), "allreduce")
        b.record()
        host = (time.perf_counter() - t0) / reps * 1e6
        torch.cuda.synchronize()
        print(f"all-reduce of {count * 8 / 1e3:.0f} KB x{reps}: host {host:.1f} us per call, "
              f"stream {a.elapsed_time(b) / reps * 1e3:.1f} us per call", flush=True)
dist.destroy_process_group()
