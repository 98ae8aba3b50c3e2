"""Per burst of a rocprofv3 --kernel-trace of `bench.py --steps K`: duration of the sweep launch, of the verdict launch
behind it, and the idle time to the next burst.    python tools/burst_trace.py DIR"""
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")))
rows.sort()
idx = [i for i, (s, e, n) in enumerate(rows) if "agent_step" in n and "true, false>" in n]
out = []
for i in idx:
    s, e, n = rows[i]
    nxt = rows[i + 1]
    if "verdict" not in nxt[2]:
        continue
    out.append(((e - s) / 1e3, (nxt[0] - e) / 1e3, (nxt[1] - nxt[0]) / 1e3, (nxt[1] - s) / 1e3))
for o in out[-12:]:
    print("sweep %7.1f us | gap %5.1f | verdict %6.1f | burst on the GPU %7.1f" % o)
if out:
    last = out[-10:]
    print("mean of the last %d: sweep %.1f verdict %.1f" % (len(last), sum(o[0] for o in last) / len(last), sum(o[2] for o in last) / len(last)))
