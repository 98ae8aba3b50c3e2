#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kadd_cold; mkdir -p $O; cd $R
timeout -k 10 600 python tests/tools/feeder_iters.py --kadd-cold > $O/kc.txt 2>&1; echo rc $?; grep "ms for 15" $O/kc.txt | cut -c1-140
