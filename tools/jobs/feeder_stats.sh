#!/bin/bash
# kernel statistics of the 15-iteration runs on the 121144 feeder (T = 24)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/feeder_stats; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f -o s -- python3 $R/tests/tools/feeder_iters.py > $O/f.log 2>&1; echo rc $?
cp $(find $O/f -name "*kernel_stats.csv" | head -1) $O/feeder_kernel_stats.csv; rm -rf $O/f; grep "ms for 15" $O/f.log | cut -c1-120; head -10 $O/feeder_kernel_stats.csv | cut -c1-150
