#!/bin/bash
# Gram kernel: rows' numbers straight into registers, the diagonal tile's rows fetched once
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ac; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 900 python -m pytest tests/test_gpu_newton.py tests/test_gpu_operator.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
for i in 1 2; do step timeout -k 10 300 python tests/tools/feeder_iters.py > $O/feeder$i.txt 2>&1; tail -1 $O/feeder$i.txt | cut -c1-160; done
step timeout -k 10 300 python tests/tools/feeder_config3.py > $O/feeder3.txt 2>&1; tail -2 $O/feeder3.txt | cut -c1-100
cd /tmp; export TMPDIR=/tmp
step timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/feeder -o s -- python3 $R/tests/tools/feeder_iters.py > $O/feeder_prof.log 2>&1
cp $(find $O/feeder -name "*kernel_stats.csv" | head -1) $O/feeder_kernel_stats.csv; rm -rf $O/feeder; head -8 $O/feeder_kernel_stats.csv | cut -c1-160
