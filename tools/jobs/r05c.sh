#!/bin/bash
# round 5, third GPU call: the whole GPU suite on the library without the sweep's closing barrier, then the bench variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1
echo "gpu tests rc $?"; tail -6 $O/gpu_tests.log
for v in "" "--pdhg-check 2" "--pdhg-check 1"; do
  timeout -k 10 300 python bench.py --steps 20 --no-extras --no-cpu-baseline --no-converge $v > "$O/bench20_${v// /_}.json" 2> $O/bench.err; echo "steps 20 $v"; python tools/show_bench.py "$O/bench20_${v// /_}.json" | head -1
  timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline --no-converge $v > "$O/bench200_${v// /_}.json" 2> $O/bench.err; echo "steps 200 $v"; python tools/show_bench.py "$O/bench200_${v// /_}.json" | head -1
done
