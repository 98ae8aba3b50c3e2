"""Could a T-double all-reduce of per-rank voltage maxima replace the M x T exchange of the node sums?
With R >= 0 and g >= 0 every rank's partial voltage v_r = R p_r is non-negative, so max_m sum_r v_r[m] <= sum_r max_m v_r[m]:
a slot whose bound is inside the limit is certified slack without the sums.  Host-side numpy on the bench workload's
coordinated profile (base load + every EV's energy spread over its window: what `stress` is scaled against, i.e. about what
the steady state settles on), residences sharded contiguously by node order as the engine shards them.
    python tools/screen_bound.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd.engine import voltage_limits
from revs_admm_amd.synthetic import make_workload

w = make_workload(100_000, 24, n_nodes=2048, seed=0, binary_feasible=False, stress=1.0)
h = w.homes
ev = h["ev"] == 1
need = np.where(ev, np.maximum(0.9 - h["initial"], 0) * h["capacity"], 0.0)
t = np.arange(24)[None, :]
inwin = ev[:, None] & (t >= h["start"][:, None]) & (t < h["end"][:, None])
peak = w.load + inwin * (need / np.maximum(h["end"] - h["start"], 1))[:, None]
_, vhi = voltage_limits(w.vset, w.vlow, w.vhigh)


def node_sum(lo, hi):
    P = np.zeros((w.M, 24))
    np.add.at(P, w.node_of[lo:hi], peak[lo:hi])
    return P


full = w.Rn @ node_sum(0, w.N)
print(f"true max voltage / limit: {full.max() / vhi:.4f}; per slot: {np.round(full.max(0) / vhi, 3).tolist()}")
for world in (2, 4, 8):
    bound = np.zeros(24)
    for r in range(world):
        bound += (w.Rn @ node_sum(*w.shard(r, world))).max(0)
    print(f"{world} ranks: sum of per-rank maxima / limit per slot: {(bound / vhi).min():.3f} .. {(bound / vhi).max():.3f}; slots certified "
          f"slack: {int((bound <= vhi).sum())} of 24; bound / true max {(bound / full.max(0)).min():.2f} .. {(bound / full.max(0)).max():.2f}")
