"""bench.py's transient and headline step for several values of OperatorOptions.newton_kadd (tuning)."""
import json
import subprocess
import sys

for k in sys.argv[1:] or ["3", "4", "6"]:
    out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--no-cpu-baseline", "--no-extras", "--no-converge",
                          "--op-kadd", k], capture_output=True, text=True).stdout
    j = json.loads(out.strip().splitlines()[-1])
    t = j["breakdown"]["transient"]
    print("kadd", k, "transient ms", round(t["ms_total"], 2), "max", round(t["ms_per_step_max"], 2), t["operator_inner_iters"][:12],
          "ms_per_step", round(j["ms_per_step"], 5), flush=True)
