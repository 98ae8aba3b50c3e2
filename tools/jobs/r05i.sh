#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
B="--no-extras --no-cpu-baseline --no-converge"
for st in 200 20 200 20; do
  timeout -k 10 300 python bench.py --steps $st $B > $O/b_$st.json 2> $O/err.txt; echo "steps $st: $(python tools/show_bench.py $O/b_$st.json 2>/dev/null | head -1)"
done
timeout -k 10 300 python bench.py --steps 20 $B --pdhg-polish 3 > $O/b_p3.json 2> $O/err.txt; echo "polish 3 steps 20: $(python tools/show_bench.py $O/b_p3.json 2>/dev/null | head -1)"
timeout -k 10 900 python -m pytest tests/test_gpu_agent.py tests/test_gpu_admm.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
