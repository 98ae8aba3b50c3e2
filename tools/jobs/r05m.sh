#!/bin/bash
# block principal pivoting started from every candidate basic (tuning build) against the product's start
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
for v in base startall; do
  if [ $v = base ]; then unset REVS_LIB; else export REVS_LIB=$R/revs_admm_amd/tune_$v.so; fi
  echo "== $v"
  step timeout -k 10 300 python tools/bpp_pivots.py > $O/piv_$v.txt 2>&1; tail -3 $O/piv_$v.txt | cut -c1-400
  step timeout -k 10 300 python tests/tools/feeder_iters.py > $O/feeder_$v.txt 2>&1; tail -1 $O/feeder_$v.txt | cut -c1-400
  step timeout -k 10 300 python tests/tools/feeder_config3.py > $O/feeder3_$v.txt 2>&1; tail -2 $O/feeder3_$v.txt | cut -c1-300
  step timeout -k 10 300 python tools/transient_times.py > $O/trans_$v.txt 2>&1; tail -4 $O/trans_$v.txt | cut -c1-300
done
