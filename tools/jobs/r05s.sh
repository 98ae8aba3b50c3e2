#!/bin/bash
# kernel timeline of the transient as one call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/tt -o tt -- python3 $R/tools/transient_trace.py 12 > $O/run.txt 2>&1; rc=$?; tail -2 $O/run.txt | cut -c1-200
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
cd $R && python tools/transient_trace.py --read $O/tt --list > $O/timeline.txt 2>&1; head -30 $O/timeline.txt
