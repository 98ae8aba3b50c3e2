#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_final; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -6 $O/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log | cut -c1-300
