#!/bin/bash
# round 5: the folded chain's sweep without its closing barrier / heaviest workgroups first (A/B), retention-based cold admission
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
for reg in binding binary; do
  for v in "old REVS_NO_WG_ORDER=1" "new REVS_NO_WG_ORDER=1" "new REVS_X=1" "old REVS_X=1" "new REVS_Y=1"; do
    lib=${v%% *}; envv=${v#* }
    echo "$reg $lib $envv: $(env REVS_LIB=$R/revs_admm_amd/tune_chain_$lib.so $envv timeout -k 10 200 python tools/regime_run.py --regime $reg 2>&1 | tail -1)"
  done
done 2>&1 | tee $O/chain_ab.txt
timeout -k 10 300 python tests/tools/feeder_iters.py > $O/feeder.txt 2>&1; tail -2 $O/feeder.txt | cut -c1-400
timeout -k 10 900 python -m pytest tests/test_gpu_admm.py tests/test_gpu_sharded.py tests/test_gpu_newton.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc $?"; python tools/show_bench.py $O/bench20.json
