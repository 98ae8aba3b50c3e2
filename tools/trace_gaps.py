"""Where the wall time of a steady-state stretch goes: the last N dispatches of a rocprofv3 --kernel-trace
directory in start order -- per kernel the mean duration, and the mean gap between one kernel's end and the next
one's start (by the pair of kernels).

    python tools/trace_gaps.py DIR [--last 400]"""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--last", type=int, default=400)
a = ap.parse_args()
rows = []
for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")))
rows.sort()
rows = rows[-a.last:]
dur, gap = defaultdict(list), defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    dur[n].append((e - s) / 1e3)
    if i + 1 < len(rows):
        gap[(n, rows[i + 1][2])].append((rows[i + 1][0] - e) / 1e3)
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"{len(rows)} dispatches over {span:.1f} us")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n[:70]:70s} x{len(v):4d}  mean {sum(v) / len(v):7.2f} us  total {sum(v):9.1f} us ({100 * sum(v) / span:4.1f} %)")
for (x, y), v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    print(f"  gap {x[:38]:38s} -> {y[:38]:38s} x{len(v):4d}  mean {sum(v) / len(v):6.2f} us  total {sum(v):8.1f} us ({100 * sum(v) / span:4.1f} %)")
