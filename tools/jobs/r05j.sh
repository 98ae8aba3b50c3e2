#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R
B="--no-extras --no-cpu-baseline --no-converge"
for inner in 32 20 16 8 32 16; do
  for st in 200 20; do
    timeout -k 10 300 python bench.py --steps $st $B --stream-inner $inner > $O/b_${inner}_$st.json 2> $O/err.txt
    echo "inner $inner steps $st: $(python tools/show_bench.py $O/b_${inner}_$st.json 2>/dev/null | head -1)"
  done
done
