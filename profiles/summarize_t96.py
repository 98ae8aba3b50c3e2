"""Counters of the T = 96 sweep (BASELINE config 4's per-GPU shape, 125 000 x 96) from the three --pmc passes of
tools/collect_profiles.sh (tools/regime_run.py --regime steady --homes 125000 --T 96; per-kernel means written by
tools/pmc_kernels.py) -> profiles/<tag>_t96_pmc_summary.csv and profiles/t96_traffic.json, which bench.py's
roofline_125k_T96 reads.

    python profiles/summarize_t96.py gpurun_out/prof_r04/pmc_t96 r04
    python profiles/summarize_t96.py gpurun_out/prof_r05/pmc_t96_1m r05 1000000      # config 4 at its whole size on one GPU
                                                                                     # -> r05_t96_1m_pmc_summary.csv, t96_1m_traffic.json
"""
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
homes, T, inner = (int(sys.argv[3]) if len(sys.argv) > 3 else 125_000), 96, 16
sfx = "" if homes == 125_000 else "_1m"
rows = []
for name in ("fetch", "write", "sq"):
    df = pd.read_csv(os.path.join(src, name, "pmc_kernels_summary.csv"))
    df.insert(0, "pass", name)
    rows.append(df)
allrows = pd.concat(rows)
allrows.to_csv(os.path.join(ROOT, "profiles", f"{tag}_t96{sfx}_pmc_summary.csv"), index=False, float_format="%.1f")
# the multi-iteration sweep: template arguments <LPA, SPL, MODE, FULL_ROWS, MULTI = true, CHAIN = false>
m = allrows[allrows.kernel.str.contains(", true, false>", regex=False)]
get = lambda p, c: float(m[m["pass"] == p][c].iloc[0])
fetch, write = get("fetch", "FETCH_SIZE") * 2 * 1024, get("write", "WRITE_SIZE") * 1024
dur = get("sq", "dur_us")
bph = 4 * 4 * T + 32 + 4 + 3 * 4 * T + 8 + 4 + 8            # bench.multi_bytes_per_home(96, "scalar")
out = {"homes": homes, "T": T, "mode": "pdhg", "iterations_per_launch": inner, "kernel": m.kernel.iloc[0],
       "avg_launch_us_profiled": dur, "fetch_bytes_corrected": fetch, "write_bytes": write,
       "hbm_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": bph * homes,
       "sq_counters_per_launch": {c: get("sq", c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_ACTIVE_INST_VALU",
                                                            "SQ_WAVE_CYCLES", "SQ_WAIT_ANY")},
       "source": f"profiles/{tag}_t96{sfx}_pmc_summary.csv (rocprofv3 --pmc, separate passes of `tools/regime_run.py --regime steady "
                 f"--homes {homes} --T 96 --steps {128 if homes == 125_000 else 64} --spin 48`; FETCH_SIZE x2 and KiB per MI355X_MICROARCH.md)"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"t96{sfx}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
