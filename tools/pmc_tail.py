"""Mean counter values of the last N dispatches of each matching kernel in a rocprofv3 --pmc directory.

    python tools/pmc_tail.py DIR [--last N] [--match agent_step]
"""
import argparse
import glob
import os

import pandas as pd

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--last", type=int, default=30)
ap.add_argument("--match", default="agent_step")
a = ap.parse_args()
for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains(a.match)]
    df["kernel"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
    for k, g in df.groupby("kernel"):
        ids = sorted(g.Dispatch_Id.unique())[-a.last:]
        t = g[g.Dispatch_Id.isin(ids)]
        dur = ((t.End_Timestamp - t.Start_Timestamp) / 1e3).mean() if "End_Timestamp" in t else float("nan")
        print(k, f"last {len(ids)} dispatches, mean {dur:.1f} us")
        for n, v in t.groupby("Counter_Name").Counter_Value.mean().items():
            print(f"    {n:28s} {v:14.1f}")
