#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/burst_cost; mkdir -p $O; cd $R
timeout -k 10 300 python tools/burst_cost.py > $O/burst_cost.txt 2>&1; echo rc $?; tail -16 $O/burst_cost.txt
