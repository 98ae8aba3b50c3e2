#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ad; mkdir -p $O; cd $R
timeout -k 10 600 python tests/tools/feeder_iters.py --nks > $O/nks.txt 2>&1; echo rc $?; grep "ms for 15" $O/nks.txt | cut -c1-110
