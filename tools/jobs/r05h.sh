#!/bin/bash
# round 5: what the multi-iteration sweep's LDS adds cost (tuning builds), and the stressed synthetic feeders (stress 6: 129+ rows?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
B="--no-extras --no-cpu-baseline --no-converge"
for lib in base nonacc presum base presum; do
  for st in 200 20; do
    REVS_LIB=$R/revs_admm_amd/tune_$lib.so timeout -k 10 300 python bench.py --steps $st $B > $O/b_${lib}_$st.json 2> $O/err.txt
    echo "$lib steps $st: $(python tools/show_bench.py $O/b_${lib}_$st.json 2>/dev/null | head -1)"
  done
done
timeout -k 10 600 python tools/stress_diag.py > $O/stress_diag.txt 2>&1; tail -12 $O/stress_diag.txt | cut -c1-600
