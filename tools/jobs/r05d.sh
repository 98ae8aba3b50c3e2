#!/bin/bash
# round 5, fourth GPU call: heavy-first workgroup order (A/B inside one job), the multi-iteration sweep at 4..8 wavefronts per SIMD,
# the bit-for-bit tests on the new library, the bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
B="--no-extras --no-cpu-baseline --no-converge"
run() { # label, env..., -- args
  local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 $B > $O/b20_$label.json 2> $O/err_$label.txt; echo "$label steps 20:  $(python tools/show_bench.py $O/b20_$label.json 2>/dev/null | head -1)"
  env "$@" timeout -k 10 300 python bench.py --steps 200 $B > $O/b200_$label.json 2>> $O/err_$label.txt; echo "$label steps 200: $(python tools/show_bench.py $O/b200_$label.json 2>/dev/null | head -1)"
}
run w5_order REVS_LIB=$R/revs_admm_amd/tune_w5.so
run w5_noorder REVS_LIB=$R/revs_admm_amd/tune_w5.so REVS_NO_WG_ORDER=1
run w5_order_again REVS_LIB=$R/revs_admm_amd/tune_w5.so
for w in 4 6 7 8; do run w$w REVS_LIB=$R/revs_admm_amd/tune_w$w.so; done
timeout -k 10 600 python -m pytest tests/test_gpu_admm.py tests/test_gpu_sharded.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc $?"; python tools/show_bench.py $O/bench20.json
