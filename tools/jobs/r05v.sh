#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
timeout -k 10 300 python tools/transient_hostgaps.py 0 > $O/hostgaps_eps.txt 2>&1; echo rc $?; head -70 $O/hostgaps_eps.txt | cut -c1-120
timeout -k 10 300 python tools/transient_hostgaps.py 15 binary > $O/hostgaps_bin.txt 2>&1; echo rc $?; head -50 $O/hostgaps_bin.txt | cut -c1-120
