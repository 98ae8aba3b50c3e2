"""Per-kernel means of a rocprofv3 --pmc pass (counter_collection.csv): one row per kernel name with the
dispatch count, the mean duration and the mean of every counter per dispatch (and per wavefront where
SQ_WAVES is among them).

    python tools/pmc_kernels.py <dir with *counter_collection.csv> [substring of the kernel names to keep]
"""
import glob
import os
import sys

import pandas as pd

src = sys.argv[1]
keep = sys.argv[2] if len(sys.argv) > 2 else "revs::"
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
df = pd.concat([pd.read_csv(f) for f in files])
df = df[df.Kernel_Name.str.contains(keep, regex=False)]
df["kernel"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
df["dur_us"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
per = df.pivot_table(index=["kernel", "Dispatch_Id", "Grid_Size"], columns="Counter_Name", values="Counter_Value",
                     aggfunc="sum").reset_index()
dur = df.groupby(["kernel", "Dispatch_Id"]).dur_us.first().reset_index()
per = per.merge(dur, on=["kernel", "Dispatch_Id"])
cols = [c for c in per.columns if c not in ("kernel", "Dispatch_Id")]
out = per.groupby(["kernel", "Grid_Size"])[[c for c in cols if c != "Grid_Size"]].mean()
out.insert(0, "dispatches", per.groupby(["kernel", "Grid_Size"]).size())
if "SQ_WAVES" in out.columns:
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if c in out.columns:
            out[c + "_per_wave"] = out[c] / out.SQ_WAVES
pd.set_option("display.width", 250, "display.max_columns", 50, "display.max_colwidth", 80)
print(out.sort_values("dur_us", ascending=False).head(12).to_string())
out.to_csv(os.path.join(src, "pmc_kernels_summary.csv"))
