#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_admm.py -m gpu -q -x -k "status_or or preallocation" > $O/tests.log 2>&1; echo "tests rc $?"; tail -15 $O/tests.log | cut -c1-200
