#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
timeout -k 10 300 python tests/tools/feeder_iters.py > $O/feeder.txt 2>&1; tail -1 $O/feeder.txt | cut -c1-400
timeout -k 10 300 python tests/tools/feeder_config3.py > $O/feeder3.txt 2>&1; tail -2 $O/feeder3.txt | cut -c1-300
timeout -k 10 900 python -m pytest tests/test_gpu_admm.py tests/test_gpu_operator.py tests/test_gpu_newton.py tests/test_gpu_config4.py -m gpu -q -x -k "not full_size" > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --no-converge > $O/bench.json 2> $O/err.txt; python tools/show_bench.py $O/bench.json | cut -c1-300
