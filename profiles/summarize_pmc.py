#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc passes of bench.py to profiles/rNN_pmc_summary.csv and
profiles/agent_traffic.json (which bench.py reports as roofline.traffic, with its source).

    python profiles/summarize_pmc.py <dir with fetch/ write/ sq/ pass directories> <round tag, e.g. r02>

Each pass directory holds <pass>_counter_collection.csv of
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir>/<pass> -o <pass> -- \
        python3 bench.py --steps 60 --no-extras --no-cpu-baseline --no-converge --clock-warm 0
Corrections, as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so it is doubled;
WRITE_SIZE is exact (and counts float atomics).  The streaming launches are told from the other
sweeps of the run by their grid (residence workgroups + T verdict workgroups)."""
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
homes, T, bytes_per_home = 100_000, 24, int(os.environ.get("REVS_BYTES_PER_HOME", "728"))
# round 3: the steady state's launches are the multi-iteration sweep (template argument MULTI = true,
# up to 32 ADMM iterations per launch -- only the full-length launches are summarised); round 2: one iteration
# per launch with T verdict workgroups in front
multi = tag >= "r03"
grid = ((homes + 31) // 32 + (0 if multi else T)) * 256
rows, res = [], {}
for name in ("fetch", "write", "sq", "sq2"):
    f = os.path.join(src, name, f"{name}_counter_collection.csv")
    if not os.path.exists(f):
        continue
    df = pd.read_csv(f)
    df = df[df.Kernel_Name.str.contains("revs::")]
    df["kernel"] = df.Kernel_Name.str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
    df["stream_launch"] = df.Kernel_Name.str.contains("agent_step") & (df.Grid_Size == grid)
    if multi:
        # template arguments <LPA, SPL, MODE, FULL_ROWS, MULTI, CHAIN>: MULTI = true, CHAIN = false
        df["stream_launch"] &= df.Kernel_Name.str.contains(", true, false>(", regex=False)
    df["dur_us"] = (df.End_Timestamp - df.Start_Timestamp) / 1e3
    if multi and df.stream_launch.any():
        # full-length launches only (a burst's remainder launch carries fewer iterations)
        full = df[df.stream_launch].dur_us.max()
        df["stream_launch"] &= df.dur_us >= 0.85 * full
    g = df.groupby(["kernel", "stream_launch", "Counter_Name"]).agg(
        calls=("Counter_Value", "size"), mean=("Counter_Value", "mean"),
        mean_dur_us=("dur_us", "mean")).reset_index()
    g.insert(0, "pass", name)
    rows.append(g)
    for n, v in df[df.stream_launch].groupby("Counter_Name").Counter_Value.mean().items():
        res[n] = float(v)
pd.concat(rows).to_csv(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.csv"), index=False, float_format="%.1f")
fetch, write = res["FETCH_SIZE"] * 2 * 1024, res["WRITE_SIZE"] * 1024
extra = {k: res[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES",
                            "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES") if k in res}
json.dump({"homes": homes, "T": T, "mode": "pdhg", "fetch_bytes_corrected": fetch, "write_bytes": write,
           "hbm_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": bytes_per_home * homes,
           "iterations_per_launch": int(os.environ.get("REVS_ITERS_PER_LAUNCH", "32")) if multi else 1,
           "sq_counters_per_launch": extra,
           "source": f"profiles/{tag}_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate "
                     "passes of `bench.py --steps 128 --warmup 32 --no-extras --no-cpu-baseline --no-converge --clock-warm 0`, "
                     "the streaming launches of each pass; counters in KiB, FETCH_SIZE x2 per "
                     "MI355X_MICROARCH.md; not measured inside the bench run)"},
          open(os.path.join(ROOT, "profiles", "agent_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
