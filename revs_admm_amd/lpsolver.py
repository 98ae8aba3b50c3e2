"""The reference's solver call surface (lpsolver.py) over the MI355X engine.

Same function names, argument meaning, return shapes and error behaviour as the
reference, so a caller of `lpsolver.solve_ADMM` / `solve_residence` /
`compute_Rmat` can switch imports:

    from revs_admm_amd.lpsolver import solve_ADMM, solve_residence, compute_Rmat

`homes` is the reference's dict {home id: {"LOAD": [...], "EV": {} | {"rating",
"capacity", "initial", "start", "end"}}} (extract.py:92-133), `graph` a networkx
graph with node attribute 'label' and edge attribute 'r' (extract.py:41-80).
Everything numeric runs in librevs_admm.so; this module only converts dicts to
arrays and back.  `grbpath` (Gurobi's log directory in the reference) is accepted
and ignored.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .engine import AdmmEngine, OperatorOptions, pack_homes, residence_solve

__all__ = ["compute_Rmat", "solve_ADMM", "solve_residence", "solve_residences",
           "solve_central", "homes_to_arrays", "feeder_arrays"]


def compute_Rmat(graph) -> np.ndarray:
    """R = 2 F D F^T over the non-substation nodes, in graph.nodes order -- the
    matrix of reference lpsolver.py:17-26.  The feeder is a tree, so instead of
    inverting the incidence matrix R[i,j] = 2 * (sum of r over the edges shared by
    the substation->i and substation->j paths) is accumulated down the tree
    (O(n^2) stores instead of an O(n^3) inverse; same numbers to 1e-18)."""
    nodes = list(graph.nodes())
    nonsub = [n for n in nodes if graph.nodes[n]["label"] != "S"]
    roots = [n for n in nodes if graph.nodes[n]["label"] == "S"]
    if len(roots) != 1 or graph.number_of_edges() != len(nodes) - 1:
        raise ValueError("compute_Rmat expects a radial feeder with one substation node")
    pos = {n: i for i, n in enumerate(nonsub)}
    R = np.zeros((len(nonsub), len(nonsub)))
    order, seen, stack = [], {roots[0]}, [roots[0]]
    parent = {}
    while stack:
        u = stack.pop()
        order.append(u)
        for v in graph.neighbors(u):
            if v not in seen:
                seen.add(v)
                parent[v] = u
                stack.append(v)
    done = []
    for v in order[1:]:
        i, p = pos[v], parent[v]
        r2 = 2.0 * graph.edges[p, v]["r"]
        if p in pos:
            row = R[pos[p], done]
            R[i, done] = row
            R[done, i] = row
            R[i, i] = R[pos[p], pos[p]] + r2
        else:
            R[i, i] = r2
        done.append(i)
    return R


def feeder_arrays(graph, res):
    """The radial feeder as (parent, edge_r, cons_of) over its non-substation nodes, for
    AdmmEngine(feeder=...): parent index (-1: fed by the substation), resistance of the edge to
    the parent, and the position in `res` of a residence node (-1 for every other node)."""
    nonsub = [n for n in graph.nodes if graph.nodes[n]["label"] != "S"]
    roots = [n for n in graph.nodes if graph.nodes[n]["label"] == "S"]
    if len(roots) != 1 or graph.number_of_edges() != graph.number_of_nodes() - 1:
        raise ValueError("feeder_arrays expects a radial feeder with one substation node")
    pos = {n: i for i, n in enumerate(nonsub)}
    parent = np.full(len(nonsub), -1, np.int64)
    edge_r = np.zeros(len(nonsub))
    seen, stack = {roots[0]}, [roots[0]]
    while stack:
        u = stack.pop()
        for v in graph.neighbors(u):
            if v not in seen:
                seen.add(v)
                parent[pos[v]] = pos.get(u, -1)
                edge_r[pos[v]] = graph.edges[u, v]["r"]
                stack.append(v)
    ridx = {h: i for i, h in enumerate(res)}
    cons_of = np.array([ridx.get(n, -1) for n in nonsub], np.int64)
    return parent, edge_r, cons_of


def homes_to_arrays(homes, res):
    """Reference `homes` dict -> (LOAD (n,T), revs_home_t records) in `res` order."""
    load = np.array([homes[h]["LOAD"] for h in res], dtype=np.float64)
    ev = np.array([homes[h]["EV"] != {} for h in res])
    get = lambda k, d: np.array([homes[h]["EV"].get(k, d) if homes[h]["EV"] else d for h in res])
    rec = pack_homes(ev, get("rating", 0.0), get("capacity", 1.0), get("initial", 0.0),
                     get("start", 0), get("end", 0))
    return load, rec


def solve_ADMM(homes, graph, cost, grbpath=None, kappa=5.0, iter_max=15, vset=1.0, vlow=0.95,
               vhigh=1.05, *, mode="binary", device="cuda:0", operator: OperatorOptions = None,
               return_engine=False):
    """Reference lpsolver.py:242-290.  Returns (diff, P_sch, S, C):
        diff[k+1][h]  float      |P_est[k+1][h] - P_sch[k+1][h]| / T
        P_sch[h]      list[T]    residence net load g_opt of the last iteration
        S[h]          list[T]    EV charger power p_opt
        C[h]          list[T+1]  state of charge s_opt
    `mode="binary"` is the reference's on/off charger (its MIQP, solved exactly);
    `mode="relaxed"` the continuous box+SOC QP solved by the batched PDHG kernel.
    A residence whose window cannot reach 90% SOC raises RevsError where the
    reference prints 'No solution found' and exits."""
    res = [n for n in graph if graph.nodes[n]["label"] == "H"]          # lpsolver.py:167
    nonsub = [n for n in graph.nodes if graph.nodes[n]["label"] != "S"]  # lpsolver.py:166
    missing = [h for h in res if h not in homes]
    if missing:
        raise KeyError(f"homes lacks residence {missing[0]} of the network")
    load, rec = homes_to_arrays(homes, res)
    if load.shape[1] != len(cost):
        raise ValueError("LOAD and cost must have the same number of slots")
    R = compute_Rmat(graph)
    pos = {n: i for i, n in enumerate(nonsub)}
    resind = [pos[n] for n in res]                                      # lpsolver.py:188-189
    R_res = R[np.ix_(resind, resind)]
    eng = AdmmEngine(np.asarray(cost, float), rec, load, np.arange(len(res)), R_res, kappa=kappa,
                     vset=vset, vlow=vlow, vhigh=vhigh, mode=mode, device=device, op=operator,
                     feeder=feeder_arrays(graph, res))
    d = eng.run(iter_max)
    P_sch, S, C = eng.result()
    diff = {k + 1: {h: float(d[k, i]) for i, h in enumerate(res)} for k in range(iter_max)}
    out = (diff, {h: P_sch[i].tolist() for i, h in enumerate(res)},
           {h: S[i].tolist() for i, h in enumerate(res)},
           {h: C[i].tolist() for i, h in enumerate(res)})
    return out + (eng,) if return_engine else out


def solve_residences(tariff, homes, device="cuda:0"):
    """Batched form of solve_residence over a `homes` dict: {h: (p, s, g)}."""
    keys = list(homes)
    load, rec = homes_to_arrays(homes, keys)
    p, s, g = residence_solve(np.asarray(tariff, float), rec, load, device)
    return {h: (p[i], s[i], g[i]) for i, h in enumerate(keys)}


def solve_residence(tariff, data, path=None):
    """Reference lpsolver.py:430-460: individual optimum of one residence
    -> (p_opt, s_opt, g_opt).  `path` (Gurobi log directory) is ignored."""
    (p, s, g), = solve_residences(tariff, {0: data}).values()
    return p, s, g


def solve_central(tariff, homes, dist, path, vset, vmin, vmax, device="cuda:0"):
    """Reference lpsolver.py:463-502: one network-wide model -- minimise sum_h tariff.g_h
    with g_h = p_h + LOAD_h, binary chargers, the SOC box (but NO s_T >= 0.9 row:
    add_home_EV, lines 338-379) and vmin^2 - vset^2 <= -R_res g[:,t] <= vmax^2 - vset^2
    (network_constraints, lines 386-405).

    Nothing in that model asks for charge, so a charger is switched on only where the price
    is negative: per residence the most negative slots of its window, at most as many as the
    SOC box allows (n_max; ties to the earlier slot) -- with every price >= 0 that is "no
    charging", which is exactly what the reference stored for the 121144 feeder (all chargers
    off, SOC 0.2).  The residences decouple as long as that schedule respects the voltage rows,
    which is checked on the GPU (revs_voltage_f32: -R_res g); then it IS the model's optimum.
    Returns (p_opt, s_opt, g_opt) dicts; raises RevsError where the reference prints
    'No solution found' (the base load alone violates a lower row -- charging only lowers
    v = -R g -- or an EV starts above 100 % SOC), and NotImplementedError where voltage rows
    couple the residences (negative prices pulling rows to their limit; an upper row violated
    by a net-exporting base load, which charging could repair): a network-wide MILP, not done."""
    import torch
    from ._lib import check, load, ptr
    from .engine import _dev_check, voltage_limits
    lib, dev = load(), _dev_check(device)
    res = [n for n in dist if dist.nodes[n]["label"] == "H"]
    nonsub = [n for n in dist.nodes if dist.nodes[n]["label"] != "S"]
    load_, rec = homes_to_arrays(homes, res)
    T, m = len(tariff), len(res)
    c = np.asarray(tariff, float)
    p = np.zeros((m, T))
    neg = np.flatnonzero(c < 0)
    if len(neg):
        order = neg[np.argsort(c[neg], kind="stable")]           # most negative first, earlier slot on ties
        for i in np.flatnonzero(rec["ev"]):
            slots = order[(order >= rec["start"][i]) & (order < rec["end"][i])][:max(int(rec["nmax"][i]), 0)]
            p[i, slots] = rec["rating"][i]
    R = compute_Rmat(dist)
    pos = {n: i for i, n in enumerate(nonsub)}
    resind = [pos[n] for n in res]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    vlo, vhi = voltage_limits(vset, vmin, vmax)
    tol = 1e-6 * max(abs(vlo), abs(vhi))
    dR = up(R[np.ix_(resind, resind)])

    def rows(g):
        """(lower rows respected, upper rows respected) for v = -R_res g"""
        dP, dV = up(g), torch.zeros(m, T, dtype=torch.float32, device=dev)
        check(lib.revs_voltage_f32(m, T, ptr(dR), ptr(dP), ptr(dV),
                                   torch.cuda.current_stream(dev).cuda_stream), "revs_voltage_f32")
        v = -dV.cpu().numpy().astype(np.float64)
        return v.min() >= vlo - tol, v.max() <= vhi + tol

    if (rec["ev"] & (rec["nmax"] < 0)).any():
        # SOC box init <= s <= 1 with init > 1 (add_home_EV, lpsolver.py:338-379): empty
        raise _lib.RevsError("No solution found (lpsolver.py:495-497): an EV's initial state of "
                             "charge is above 1")
    lo_ok, hi_ok = rows(load_ + p)
    if not (lo_ok and hi_ok):
        base_lo, base_hi = rows(load_) if p.any() else (lo_ok, hi_ok)
        # Charging (p >= 0, R >= 0) only lowers v = -R g: a lower row the base load already
        # violates cannot be repaired -- the reference's 'No solution found'.  Everything else
        # (negative prices pulling rows to their limit; an upper row violated by a net-exporting
        # base load, which charging could repair) is a coupled network-wide MILP.
        if not base_lo:
            raise _lib.RevsError("No solution found (lpsolver.py:495-497): the base load alone "
                                 "violates the lower voltage limit")
        raise NotImplementedError("solve_central: the voltage rows bind (negative prices, or a base load "
                                  "above the upper limit that charging could repair) and the residences "
                                  "no longer decouple (DESIGN.md section 7)")
    cap = np.where(rec["ev"], rec["capacity"], 1.0)
    soc = rec["initial"][:, None] + np.concatenate([np.zeros((m, 1)), np.cumsum(p, 1)], 1) / cap[:, None]
    p_opt = {h: p[i].copy() for i, h in enumerate(res)}
    s_opt = {h: (soc[i] if rec["ev"][i] else np.zeros(T + 1)) for i, h in enumerate(res)}
    g_opt = {h: load_[i] + p[i] for i, h in enumerate(res)}
    return p_opt, s_opt, g_opt
