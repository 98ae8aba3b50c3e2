#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_newton.py -m gpu -q -s -k "big or more_than_128 or model_problem_sizes" > $O/big_tests.log 2>&1; echo "big tests rc $?"; grep -v "^$" $O/big_tests.log | tail -25 | cut -c1-400
timeout -k 10 300 python tests/tools/feeder_iters.py > $O/feeder.txt 2>&1; tail -1 $O/feeder.txt | cut -c1-400
timeout -k 10 300 python tests/tools/feeder_config3.py > $O/feeder3.txt 2>&1; tail -3 $O/feeder3.txt | cut -c1-300
