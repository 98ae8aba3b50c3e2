"""Digest of a bench.py JSON line: python tools/show_bench.py file.json"""
import json
import sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step", round(j["ms_per_step"], 5), "value", f"{j['value']:.4g}", "bursts", [round(x, 5) for x in j.get("bursts_ms_per_step", [])])
for k in ("value_binding", "value_binary", "value_125k_T96", "value_1M_T96", "value_kkt_presolve"):
    if j.get(k):
        print(k, round(j[k]["ms_per_step"], 5), [round(x, 5) for x in j[k]["blocks_ms_per_step"]])
if j.get("value_feeder_121144"):
    for tag, r in j["value_feeder_121144"].items():
        print(tag, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k != "newton_iterations"})
print("roofline", {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms")})
if j.get("roofline_valu"):
    print("valu", {k: j["roofline_valu"][k] for k in ("frac", "valu_instructions_per_wavefront_and_iteration")})
if j.get("cpu_baseline"):
    print("cpu", round(j["cpu_baseline"]["value"]), j["cpu_baseline"]["cores"])
t = j["breakdown"]["transient"]
print("transient", round(t["ms_total"], 2), round(t["ms_per_step_mean"], 3), round(t["ms_per_step_max"], 2), "one call", t.get("one_call_ms_total") and round(t["one_call_ms_total"], 2), "iters_to_eps", j["breakdown"]["admm_iters_to_eps"])
if j.get("time_to_eps"):
    te = j["time_to_eps"]
    print("time_to_eps_ms", te["time_to_eps_ms"] and round(te["time_to_eps_ms"], 3), "iterations", te["iterations"], "runs", [round(x, 2) for x in te["runs_ms"] or []],
          "with history", te["with_per_residence_diff_history_ms"] and round(te["with_per_residence_diff_history_ms"], 2), "cpu projected s", te["cpu_time_to_eps_s_projected"])
    if te.get("kkt_presolve"):
        print("time_to_eps with the KKT steps before PDHG ms", round(te["kkt_presolve"]["ms"], 3), "iterations", te["kkt_presolve"]["iterations"], [round(x, 2) for x in te["kkt_presolve"]["runs_ms"]])
    rr = te["reference_rule"]
    print("reference rule (15 iterations, binary) ms", round(rr["ms"], 3), [round(x, 2) for x in rr["runs_ms"]], "with history", round(rr["with_per_residence_diff_history_ms"], 2))
for k in ("roofline_matvec", "roofline_matvec_config3", "roofline_1M_T96"):
    if j.get(k):
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in j[k].items() if a in ("achieved", "frac", "avg_launch_ms", "M", "T", "ms_per_step_live")})
