#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t; mkdir -p $O; cd $R
timeout -k 10 300 python tools/transient_hostgaps.py 12 > $O/hostgaps.txt 2>&1; echo rc $?; head -80 $O/hostgaps.txt | cut -c1-160
