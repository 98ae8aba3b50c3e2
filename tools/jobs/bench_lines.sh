#!/bin/bash
# final bench lines (default command and the driver's), timed
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/bench_lines; mkdir -p $O; cd $R
s=$(date +%s); timeout -k 10 500 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "default bench rc $? in $(( $(date +%s) - s )) s"
s=$(date +%s); timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_steps20.json 2> $O/bench20.err; echo "driver bench rc $? in $(( $(date +%s) - s )) s"
python tools/show_bench.py $O/bench_line_driver_steps20.json | cut -c1-200 | head -12
python tools/show_bench.py $O/bench_line.json | cut -c1-200 | head -3
