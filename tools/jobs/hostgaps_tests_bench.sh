#!/bin/bash
# host time between the native calls of a run to the eps-residual, the host cost of a burst of 20, part of the suite, a bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hostgaps; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python tools/transient_hostgaps.py 0 > $O/hostgaps_eps.txt 2>&1; head -4 $O/hostgaps_eps.txt; tail -1 $O/hostgaps_eps.txt
step timeout -k 10 300 python tools/burst_cost.py > $O/burst_cost.txt 2>&1; tail -1 $O/burst_cost.txt
step timeout -k 10 1000 python -m pytest tests/test_gpu_admm.py tests/test_gpu_config4.py tests/test_gpu_sharded.py tests/test_abi.py -m gpu -q -x -k "not full_size" > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
step timeout -k 10 400 python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/err.txt; python tools/show_bench.py $O/bench.json | grep -i "eps\|reference rule\|transient\|ms_per_step"
