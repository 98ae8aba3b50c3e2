#!/bin/bash
# kernel statistics of BASELINE config 3 (all communities of the 121144 feeder, T = 96): 15 iterations, on/off chargers and PDHG
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/config3_stats; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o s -- python3 $R/tests/tools/feeder_config3.py > $O/c3.log 2>&1; echo rc $?
cp $(find $O/c3 -name "*kernel_stats.csv" | head -1) $O/config3_kernel_stats.csv; rm -rf $O/c3; tail -2 $O/c3.log | cut -c1-120; head -12 $O/config3_kernel_stats.csv | cut -c1-150
