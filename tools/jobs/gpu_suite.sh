#!/bin/bash
# the whole GPU suite + smoke
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/gpu_suite; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
