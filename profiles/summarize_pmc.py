#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes (gpurun_out/pmc*_{FETCH,WRITE}_SIZE) to a per-kernel
summary and to profiles/agent_traffic.json, which bench.py reports as roofline.traffic.

Corrections, as MI355X_MICROARCH.md section HBM prescribes: counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read,
so it is doubled; WRITE_SIZE is exact.  The agent kernel reads every input element
exactly once by construction (no reuse), so its own algorithmic read bytes are the
calibration point for its access pattern (12 B per lane): 2 x FETCH_SIZE x 1024
= 1.002 x the algorithmic reads at 100k homes and 1.000 x at 1M homes."""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES_PER_HOME = int(os.environ.get("REVS_BYTES_PER_HOME", "824"))   # bench.py agent_bytes_per_home
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
rows = []
for tag, homes in (("pmc", 100000), ("pmc1m", 1000000)):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"{src}/{tag}_{ctr}/**/*counter_collection.csv", recursive=True)
        if not fs:
            continue
        df = pd.read_csv(max(fs, key=os.path.getmtime))      # newest pass
        df = df[df["Kernel_Name"].str.contains("revs::")]
        df["kernel"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
        df["dur_us"] = (df["End_Timestamp"] - df["Start_Timestamp"]) / 1e3
        g = df.groupby("kernel").agg(calls=("Counter_Value", "size"), mean_KiB=("Counter_Value", "mean"),
                                     mean_dur_us=("dur_us", "mean")).reset_index()
        g.insert(0, "counter", ctr)
        g.insert(0, "homes", homes)
        rows.append(g)
out = pd.concat(rows)
out["bytes_corrected"] = out["mean_KiB"] * 1024 * out["counter"].map({"FETCH_SIZE": 2.0, "WRITE_SIZE": 1.0})
out.to_csv(os.path.join(ROOT, "profiles", "r01_pmc_summary.csv"), index=False, float_format="%.1f")
a = out[(out.homes == 100000) & out.kernel.str.contains("agent_step_kernel")]
fetch = float(a[a.counter == "FETCH_SIZE"].bytes_corrected.iloc[0])
write = float(a[a.counter == "WRITE_SIZE"].bytes_corrected.iloc[0])
json.dump({"homes": 100000, "T": 24, "mode": "pdhg", "fetch_bytes_corrected": fetch,
           "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
           "algorithmic_bytes_per_launch": BYTES_PER_HOME * 100000,
           "source": "profiles/r01_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, "
                     "separate passes; FETCH_SIZE x2 per MI355X_MICROARCH.md)"},
          open(os.path.join(ROOT, "profiles", "agent_traffic.json"), "w"), indent=1)
print(out.to_string())
