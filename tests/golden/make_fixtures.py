#!/usr/bin/env python3
"""Generate tests/golden/revs_121144.npz from the reference's DATA files.

Run once, in the build container (the GPU box has no /root/reference):

    python tests/golden/make_fixtures.py

Only data is read -- the reference's input files (`input/121144-dist-net.gpickle`,
`input/121144-com.txt`, `input/DVP-tariff.txt`) and the result files the
reference itself stored under `out/121144-com2/{individual,centralized,
distributed}/` (written by revs_fixture.py:216-220, 243-247, 274-278 through
extract.py:combine_result).  No reference code is imported or executed: the
reference needs gurobipy, which this image does not have.

What ends up in the fixture
---------------------------
* the feeder as plain arrays (node ids/labels, edge endpoints, edge resistance)
  in the reference's own node/edge order, so compute_Rmat (lpsolver.py:17-26)
  can be restated on it;
* the shifted tariff (extract.py:16-24, shift=6) and the five communities;
* the hourly base LOAD of every residence.  The reference's load CSV
  (`121-home-load.csv`) is not in the repository, but every stored result holds
  g = p + LOAD (lpsolver.py:64-65) and p for the EV homes, so LOAD = g - p; the
  three stored result sets agree on it to 1e-12, which this script asserts;
* the reference's stored answers: individual (3 cases), centralized and
  distributed (P_res, P_ev, SOC, and the per-iteration `diff` trajectory of
  lpsolver.py:284).

The network pickle holds shapely geometries (edge attribute `geometry`); they
are irrelevant to the electrical model, so the unpickler maps shapely classes
to an inert placeholder instead of requiring shapely.
"""
import os
import pickle
import sys

import numpy as np

REF = os.environ.get("REVS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        pass


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("shapely"):
            return _Inert
        return super().find_class(module, name)


def read_result(path):
    """Parse a combine_result() text file (extract.py:135-174) into
    {section title: {home id: np.ndarray}}."""
    lines = open(path).read().split("\n")
    out, cur, i = {}, None, 0
    while i < len(lines):
        if (lines[i].startswith("####") and i + 2 < len(lines)
                and lines[i + 2].startswith("####")):
            cur = lines[i + 1]
            out[cur] = {}
            i += 3
            continue
        if lines[i].strip() and cur is not None:
            h, v = lines[i].split(":\t")
            out[cur][int(h)] = np.array([float(x) for x in v.split(" ")])
        i += 1
    return out


def main():
    g = _Unpickler(open(f"{REF}/input/121144-dist-net.gpickle", "rb")).load()
    nodes = list(g.nodes())
    edges = list(g.edges())
    nidx = {n: i for i, n in enumerate(nodes)}
    fx = {
        "node_id": np.array(nodes, dtype=np.int64),
        "node_label": np.array([g.nodes[n]["label"] for n in nodes], dtype="S1"),
        "edge_u": np.array([nidx[u] for u, v in edges], dtype=np.int32),
        "edge_v": np.array([nidx[v] for u, v in edges], dtype=np.int32),
        "edge_r": np.array([g.edges[e]["r"] for e in edges], dtype=np.float64),
    }
    res = [n for n in nodes if g.nodes[n]["label"] == "H"]
    fx["res_id"] = np.array(res, dtype=np.int64)

    with open(f"{REF}/input/DVP-tariff.txt") as f:
        tariff = [float(x) for x in f.readline().split(" ")]
    fx["tariff_raw"] = np.array(tariff)
    fx["tariff_shift6"] = np.roll(tariff, -6)

    com_lines = open(f"{REF}/input/121144-com.txt").readlines()
    coms = [[int(x) for x in l.strip("\n").split(" ")] for l in com_lines]
    fx["com_flat"] = np.array(sum(coms, []), dtype=np.int64)
    fx["com_offsets"] = np.cumsum([0] + [len(c) for c in coms]).astype(np.int64)

    K_RES = "Residence Usage Profile"
    K_EV = "EV Charger Usage Profile"
    K_SOC = "EV Charger State of Charge Profile"
    K_DIFF = "EV Convergence over Iterations"
    cases = {
        "ind_a90_r4800": "individual/adopt90-rating4800-seed1234.txt",
        "ind_a70_r4800": "individual/adopt70-rating4800-seed1234.txt",
        "ind_a90_r3600": "individual/adopt90-rating3600-seed1234.txt",
        "cen_a90_r4800": "centralized/adopt90-rating4800-seed1234.txt",
        "dis_a90_r4800": "distributed/adopt90-rating4800-seed1234.txt",
    }
    load = None
    for tag, rel in cases.items():
        s = read_result(f"{REF}/out/121144-com2/{rel}")
        assert list(s[K_RES]) == res, "result rows follow the graph's home order"
        ev = list(s[K_EV])
        fx[f"{tag}_ev_homes"] = np.array(ev, dtype=np.int64)
        fx[f"{tag}_P_res"] = np.array([s[K_RES][h] for h in res])
        fx[f"{tag}_P_ev"] = np.array([s[K_EV][h] for h in ev])
        fx[f"{tag}_SOC"] = np.array([s[K_SOC][h] for h in ev])
        if K_DIFF in s:
            fx[f"{tag}_diff"] = np.array([s[K_DIFF][h] for h in ev])
        this = fx[f"{tag}_P_res"].copy()
        for j, h in enumerate(ev):
            this[res.index(h)] -= fx[f"{tag}_P_ev"][j]
        if load is None:
            load = this
        assert np.abs(this - load).max() < 1e-12, tag
        # the reference draws EV homes with numpy's legacy generator
        # (revs_fixture.py:174-177); confirm the stored order is that draw
        adopt = int(tag.split("_a")[1][:2])
        np.random.seed(1234)
        draw = np.random.choice(coms[1], int(adopt * 1e-2 * len(coms[1])), replace=False)
        assert list(draw) == ev, tag
    fx["LOAD"] = load

    out = os.path.join(HERE, "revs_121144.npz")
    np.savez_compressed(out, **fx)
    print(f"wrote {out}: {os.path.getsize(out)/1e6:.2f} MB, "
          f"{len(nodes)} nodes, {len(res)} residences")


if __name__ == "__main__":
    sys.exit(main())
