#!/bin/bash
# round 5, first GPU call: probes (raw output for profiles/), the new and tightened tests, then the whole GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
for p in valu_rate wave_reduce column_probe icache_probe; do
  timeout -k 10 120 tools/probes/$p.bin > $O/probe_$p.txt 2>&1 || echo "probe $p rc $?" >> $O/probe_$p.txt
done
echo probes done
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "full_size or eight_logical or binary or config3 or config0 or relaxed_trajectory or edge or individual or residence" > $O/new_tests.log 2>&1
echo "new tests rc $?"; tail -5 $O/new_tests.log
