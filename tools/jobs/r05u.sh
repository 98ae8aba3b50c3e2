#!/bin/bash
# buffers allocated at construction: host gaps of the transient, the GPU suite, the bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05u; mkdir -p $O; cd $R
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: $*"; exit $rc; fi; return $rc; }
step timeout -k 10 300 python tools/transient_hostgaps.py 12 > $O/hostgaps.txt 2>&1; head -40 $O/hostgaps.txt | cut -c1-120
step timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
step timeout -k 10 400 python bench.py --steps 20 > $O/bench.json 2> $O/err.txt; python tools/show_bench.py $O/bench.json | cut -c1-300
