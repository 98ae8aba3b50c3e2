#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/config3_nks; mkdir -p $O; cd $R
for n in 0 2 3 4 6; do
  timeout -k 10 300 python tests/tools/feeder_config3.py --nks $n > $O/nks_$n.txt 2>&1; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  echo "nks $n"; tail -2 $O/nks_$n.txt | cut -c1-70
done
