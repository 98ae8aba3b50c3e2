#!/bin/bash
# stage stamps of the pivoting kernel (tuning build revs_admm_amd/tune_bpp.so), Newton tests, the feeder
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bpp_stamps; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
REVS_LIB=$R/revs_admm_amd/tune_bpp.so step timeout -k 10 300 python tools/bpp_stamps.py > $O/stamps.txt 2>&1; tail -4 $O/stamps.txt | cut -c1-200
step timeout -k 10 900 python -m pytest tests/test_gpu_newton.py tests/test_gpu_operator.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
for i in 1 2 3; do step timeout -k 10 300 python tests/tools/feeder_iters.py > $O/feeder$i.txt 2>&1; tail -1 $O/feeder$i.txt | cut -c1-100; done
