#!/bin/bash
# stage stamps of the pivoting kernel: committed form against the working tree's
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
for v in bpp_old bpp; do
  echo "== $v"
  REVS_LIB=$R/revs_admm_amd/tune_$v.so step timeout -k 10 300 python tools/bpp_stamps.py > $O/stamps_$v.txt 2>&1; tail -4 $O/stamps_$v.txt | cut -c1-300
done
