#!/bin/bash
# bash tools/jobs/one_test.sh is run as: gpurun -- 'K="expr" F="tests/file.py" bash tools/jobs/one_test.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/one_test; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest ${F:-tests} -m gpu -q -x -k "${K:-gpu}" > $O/tests.log 2>&1; echo "tests rc $?"; tail -25 $O/tests.log | cut -c1-220
