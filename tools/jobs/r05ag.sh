#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ag; mkdir -p $O; cd $R
REVS_LIB=$R/revs_admm_amd/tune_rows.so timeout -k 10 300 python tools/rows_stamps.py > $O/rows_feeder.txt 2>&1; echo rc $?; tail -15 $O/rows_feeder.txt
REVS_LIB=$R/revs_admm_amd/tune_rows.so timeout -k 10 300 python tools/rows_stamps.py --synthetic > $O/rows_syn.txt 2>&1; echo rc $?; tail -15 $O/rows_syn.txt
