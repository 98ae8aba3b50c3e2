#!/bin/bash
# the N > 1 path of bench.py rehearsed on ONE GPU (--share-gpu: all ranks on cuda:0, node sums through the library's hook
# communicator over gloo) with 2 and 4 ranks, launched both ways the contract knows: by bench.py itself and by
# torch.distributed.run.  Not a performance figure: the ranks share the card.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/rehearse_ranks; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 400 python bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 > $O/n2_self.json 2> $O/n2_self.err; echo "n2 self rc $?"; tail -c 600 $O/n2_self.json | head -c 600; echo
step timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --share-gpu --steps 20 --warmup 5 > $O/n4_torchrun.json 2> $O/n4_torchrun.err; echo "n4 torchrun rc $?"
python - <<'PY'
import json
for f in ("n2_self", "n4_torchrun"):
    try:
        j = json.loads(open(f"gpurun_out/rehearse_ranks/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: j[k] for k in ("n_gpus", "value", "ms_per_step", "scaling")}, j["config"]["parallelism"][:80], j.get("collective"))
    except Exception as e:
        print(f, "no line:", e)
PY
