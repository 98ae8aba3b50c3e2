#!/bin/bash
# shifts of the evaluations by the tree form: Newton / operator / ADMM tests, the feeder at T = 24 and T = 96
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/shift_tree; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 1000 python -m pytest tests/test_gpu_newton.py tests/test_gpu_operator.py tests/test_gpu_admm.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for i in 1 2; do step timeout -k 10 300 python tests/tools/feeder_iters.py > $O/feeder$i.txt 2>&1; tail -1 $O/feeder$i.txt | cut -c1-160; done
step timeout -k 10 300 python tests/tools/feeder_config3.py > $O/feeder3.txt 2>&1; tail -2 $O/feeder3.txt | cut -c1-100
