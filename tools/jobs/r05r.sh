#!/bin/bash
# column slabs of the model Hessian: 8 (M // 128) against 4 / 16 / 32 on the reference's feeder; Newton tests with the working tree's kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05r; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
for v in 8 4 16 32; do
  echo "== nks $v"
  REVS_NKS_X=$v step timeout -k 10 300 python tests/tools/feeder_iters.py > $O/feeder_$v.txt 2>&1; tail -1 $O/feeder_$v.txt | cut -c1-120
  REVS_NKS_X=$v step timeout -k 10 300 python tests/tools/feeder_config3.py > $O/feeder3_$v.txt 2>&1; tail -2 $O/feeder3_$v.txt | cut -c1-80
done
step timeout -k 10 900 python -m pytest tests/test_gpu_newton.py tests/test_gpu_operator.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
