"""Synthetic N-home x T-slot workloads (BASELINE.json configs 1, 2, 4).

Host-side numpy only: builds inputs, computes nothing of the hot path.  The
shapes follow the reference's data model: a radial feeder whose residences hang
off service nodes (extract.py:41-80), hourly base load per residence
(extract.py:26-44), a time-of-use tariff (extract.py:16-24) and per-residence EV
parameters (extract.py:92-133).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .engine import pack_homes, voltage_limits

DVP_TARIFF = np.array([0.07866] * 5 + [0.095111] * 10 + [0.214357] * 3 + [0.095111] * 6)


@dataclass
class Workload:
    cost: np.ndarray        # (T,)
    load: np.ndarray        # (N,T) float64
    homes: np.ndarray       # (N,) HOME_DTYPE
    node_of: np.ndarray     # (N,) int64
    Rn: np.ndarray          # (M,M) float64
    parent: np.ndarray      # (M,) parent node (-1 = substation)
    edge_r: np.ndarray      # (M,) resistance of the edge to the parent
    vset: float
    vlow: float
    vhigh: float
    kappa: float = 5.0

    @property
    def N(self):
        return self.load.shape[0]

    @property
    def T(self):
        return self.load.shape[1]

    @property
    def M(self):
        return self.Rn.shape[0]

    @property
    def feeder(self):
        """(parent, edge_r, cons_of) for AdmmEngine(feeder=...): every tree node is a constraint node."""
        return self.parent, self.edge_r, np.arange(self.M)

    def shard(self, rank, world):
        """Contiguous block of residences for one rank (nodes stay replicated)."""
        lo, hi = (self.N * rank) // world, (self.N * (rank + 1)) // world
        return lo, hi


def radial_R(parent, edge_r):
    """LinDistFlow matrix of a rooted tree: R[i,j] = 2 * sum of r over the edges
    common to the root->i and root->j paths (= 2 F D F^T of lpsolver.py:17-26).
    parent[i] < i for every node (topological order)."""
    M = len(parent)
    R = np.zeros((M, M))
    for i in range(M):
        p = parent[i]
        if p >= 0:
            R[i, :i] = R[p, :i]
            R[:i, i] = R[i, :i]
            R[i, i] = R[p, p] + 2.0 * edge_r[i]
        else:
            R[i, i] = 2.0 * edge_r[i]
    return R


def make_workload(n_homes, T=24, n_nodes=None, seed=0, adoption=0.5, binary_feasible=True,
                  stress=1.0, vset=1.0, vlow=0.95, vhigh=1.05, kappa=5.0) -> Workload:
    """Random feeder + residences.  `stress` scales the line resistances so that the
    coordinated profile (base load + every EV's energy spread evenly over its window) sits
    at stress x the upper voltage limit at the worst node: with stress > 1 the operator
    constraint binds on a minority of nodes and slots in the ADMM's steady state, and much
    harder in its first iterations, when every charger picks the cheapest slots."""
    rng = np.random.default_rng(seed)
    M = n_nodes or int(np.clip(n_homes // 8, 8, 2048))
    # feeder: a few long laterals with short branches
    parent = np.full(M, -1, np.int64)
    for i in range(1, M):
        lo = max(0, i - 12)
        parent[i] = rng.integers(lo, i) if rng.random() < 0.9 else rng.integers(0, i)
    n_feed = max(1, M // 64)
    parent[:n_feed] = -1
    edge_r = rng.uniform(0.5, 1.5, M)
    R = radial_R(parent, edge_r)
    node_of = np.sort(rng.integers(0, M, n_homes)).astype(np.int64)

    hours = (np.arange(T) + 0.5) * 24.0 / T
    shape = (0.45 + 0.25 * np.exp(-0.5 * ((hours - 2.0) / 2.5) ** 2)      # shifted day:
             + 0.55 * np.exp(-0.5 * ((hours - 13.0) / 3.0) ** 2))         # slot 0 = 06:00
    base = rng.lognormal(mean=0.0, sigma=0.35, size=(n_homes, 1))
    load = base * shape[None, :] * rng.uniform(0.85, 1.15, (n_homes, T))
    cost = np.roll(np.repeat(DVP_TARIFF, max(1, T // 24))[:T] if T >= 24
                   else DVP_TARIFF[:T], -6 * max(1, T // 24))
    if len(cost) < T:
        cost = np.resize(cost, T)

    slot_h = 24.0 / T
    ev = rng.random(n_homes) < adoption
    rating = rng.choice([3.6, 4.8, 7.2], n_homes)
    cap_kwh = rng.choice([20.0, 40.0, 60.0], n_homes)
    capacity = cap_kwh / slot_h                    # kW-slots: s += p / capacity per slot
    start = (rng.integers(10, 14, n_homes) * T) // 24
    end = np.minimum((rng.integers(21, 25, n_homes) * T) // 24, T)
    per = rating / capacity
    window = end - start
    if binary_feasible:
        kmax = np.minimum(np.floor(0.8 / per), window - 1).astype(np.int64)
        kmax = np.maximum(kmax, 1)
        k = (1 + np.floor(rng.random(n_homes) * kmax)).astype(np.int64)
        k = np.minimum(k, kmax)
        initial = 0.9 + 0.1 * rng.uniform(0.15, 0.85, n_homes) - k * per
        bad = initial < 0.02
        initial = np.where(bad, 0.9 + 0.05 - 1 * per, initial)
        initial = np.maximum(initial, 0.0)
    else:
        need = rng.uniform(0.3, 0.7, n_homes)
        initial = np.clip(0.9 - need, 0.05, 0.85)
        # keep the energy reachable inside the window
        reach = per * (window - 1)
        initial = np.maximum(initial, 0.9 - 0.9 * reach)
    homes = pack_homes(ev, rating, capacity, initial, start, end)

    # scale resistances against the COORDINATED profile: every EV's energy need spread evenly
    # over its window on top of the base load (about what the relaxed ADMM settles on).  The
    # price-driven schedules of the first ADMM iterations peak well above it.
    need = np.where(ev, np.maximum(0.9 - initial, 0.0) * capacity, 0.0)       # kW-slots
    t = np.arange(T)[None, :]
    inwin = ev[:, None] & (t >= start[:, None]) & (t < end[:, None])
    peak = load + inwin * (need / np.maximum(window, 1))[:, None]
    Pn = np.zeros((M, T))
    np.add.at(Pn, node_of, peak)
    _, vhi = voltage_limits(vset, vlow, vhigh)
    scale = stress * vhi / (R @ Pn).max()
    return Workload(cost, load, homes, node_of, R * scale, parent, edge_r * scale, vset, vlow,
                    vhigh, kappa)
