#!/bin/bash
# gpurun_out/prof_$TAG (tools/collect_profiles.sh parts A and B) -> the summaries under profiles/ and DESIGN.md's table.
#     bash tools/assemble_profiles.sh r05
set -e
TAG=${1:-r05}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_$TAG
P=$R/profiles
cp $O/bench_line.json $P/${TAG}_bench_line.json
cp $O/bench_line_driver_steps20.json $P/${TAG}_bench_line_driver_steps20.json
for f in bench binding binary feeder; do cp $O/${f}_kernel_stats.csv $P/${TAG}_${f}_kernel_stats.csv; done
cp $O/burst_trace_steps20.txt $P/${TAG}_burst_trace_steps20.txt
python3 $P/summarize_pmc.py $O/pmc $TAG
python3 $P/summarize_t96.py $O/pmc_t96 $TAG
python3 $P/summarize_t96.py $O/pmc_t96_1m $TAG 1000000
python3 - "$O" "$P/${TAG}_pmc_mfma.csv" <<'PY'
import sys
o, out = sys.argv[1], sys.argv[2]
rows = []
for sn, label in (("syn", "M = 2048 T = 24 (synthetic feeder)"), ("config3", "M = 1126 T = 96 (BASELINE config 3: 121144 feeder)")):
    lines = open(f"{o}/pmc_mfma_{sn}.csv").read().strip().splitlines()
    if not rows:
        rows.append("shape," + lines[0])
    rows += [f'"{label}",' + l for l in lines[1:]]
open(out, "w").write("\n".join(rows) + "\n")
PY
python3 $R/tools/design_table.py $TAG > /dev/null
echo "profiles/${TAG}_* written, DESIGN.md section 6 regenerated"
