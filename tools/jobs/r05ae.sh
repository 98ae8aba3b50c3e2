#!/bin/bash
# shifts from the listed rows of R up to 48 / 80 / 128 rows per slot (beyond: the dense product)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05af; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
for v in 48 32 16; do
  echo "== few $v"
  if [ $v = 48 ]; then unset REVS_LIB; else export REVS_LIB=$R/revs_admm_amd/tune_few$v.so; fi
  export REVS_DUAL_FEW_X=$v
  for i in 1 2; do step timeout -k 10 300 python tests/tools/feeder_iters.py > $O/feeder_$v.txt 2>&1; tail -1 $O/feeder_$v.txt | cut -c1-120; done
  step timeout -k 10 300 python tests/tools/feeder_config3.py > $O/feeder3_$v.txt 2>&1; tail -2 $O/feeder3_$v.txt | cut -c1-80
done
