"""Durations of the last launches of each kernel in a rocprofv3 --kernel-trace directory.

    python tools/trace_tail.py DIR [--groups N --size S]

Prints, per kernel name (shortened), the mean / min duration of consecutive groups of S dispatches
counted from the end (tools/multi_probe.py times kin = 1..4 in groups of --reps launches)."""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--groups", type=int, default=3)
ap.add_argument("--size", type=int, default=50)
ap.add_argument("--match", default="agent_step")
a = ap.parse_args()
rows = defaultdict(list)
for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if a.match in r["Kernel_Name"]:
            rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for name, v in rows.items():
    v.sort()
    short = re.sub(r"\(.*", "", name)
    print(short, len(v), "dispatches")
    for g in range(a.groups):
        seg = v[len(v) - (g + 1) * a.size: len(v) - g * a.size]
        if len(seg) < a.size:
            break
        d = [(e - s) / 1e3 for s, e in seg]
        gaps = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(len(seg) - 1)]
        print(f"  group -{g + 1}: mean {sum(d) / len(d):.2f} us  min {min(d):.2f}  max {max(d):.2f}  "
              f"mean gap to next {sum(gaps) / max(len(gaps), 1):.2f} us")
