#!/bin/bash
# round 5, second GPU call: VALU probe (span-based), the new / tightened tests (all of them, no -x), the feeder with the
# cold admission rule, the transient per iteration, the bench line, the 4 x 6 lane shape
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
timeout -k 10 120 tools/probes/valu_rate.bin > $O/probe_valu_rate.txt 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -q -s -k "full_size or eight_logical or binary or config3 or config0 or relaxed_trajectory or edge or individual or residence or tiny" > $O/new_tests.log 2>&1
echo "new tests rc $?"; tail -8 $O/new_tests.log
timeout -k 10 300 python tests/tools/feeder_iters.py --kadd-cold > $O/feeder_kadd_cold.txt 2>&1; tail -12 $O/feeder_kadd_cold.txt
timeout -k 10 200 python tools/transient_times.py > $O/transient_times.txt 2>&1; tail -40 $O/transient_times.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc $?"; python tools/show_bench.py $O/bench20.json
REVS_AGENT_SHAPE=4x6 timeout -k 10 300 python bench.py --steps 20 --no-extras --no-cpu-baseline --no-converge > $O/bench20_4x6.json 2> $O/bench20_4x6.err; python tools/show_bench.py $O/bench20_4x6.json
REVS_AGENT_SHAPE=4x6 timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline --no-converge > $O/bench200_4x6.json 2> $O/bench200_4x6.err; python tools/show_bench.py $O/bench200_4x6.json
timeout -k 10 300 python bench.py --steps 200 --no-extras --no-cpu-baseline --no-converge > $O/bench200.json 2> $O/bench200.err; python tools/show_bench.py $O/bench200.json
