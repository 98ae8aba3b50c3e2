#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
timeout -k 10 600 python tools/newton_trace.py > $O/newton_trace.txt 2>&1; echo "rc $?"; tail -5 $O/newton_trace.txt | cut -c1-1500
