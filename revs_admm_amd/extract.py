"""Input readers and result writer with the reference's names (extract.py).

Host-side I/O only (row (f) of DESIGN.md: the callers and data formats either side
of the hot path)."""
from __future__ import annotations

import os
import pickle

import numpy as np


def GetTariff(path, region, shift):
    """extract.py:16-24: one line of T prices, rolled left by `shift`."""
    f = f"{path}/{region}-tariff.txt"
    if not os.path.exists(f):
        raise ValueError(f"{f} doesn't exist!")
    with open(f) as fh:
        tariff = [float(x) for x in fh.readline().split(" ")]
    return np.roll(tariff, -shift).tolist()


def GetHomeLoad(path, region_list, shift):
    """extract.py:26-44: {hid: 24 hourly kW values rolled by shift} from
    `<region>-home-load.csv` (columns hid, hour1..hour24 in W)."""
    import pandas as pd
    home_data = {}
    if not isinstance(region_list, (list, tuple)):
        region_list = [region_list]
    for reg in region_list:
        f = f"{path}/{reg}-home-load.csv"
        if not os.path.exists(f):
            raise ValueError(f"{f} doesn't exist!")
        df = pd.read_csv(f).set_index("hid")
        cols = [f"hour{i + 1}" for i in range(24)]
        vals = 1e-3 * df[cols].to_numpy(float)
        for h, row in zip(df.index, vals):
            home_data[h] = np.roll(row, -shift).tolist()
    return home_data


class _Geometry:
    """Placeholder for shapely geometries stored on the edges of the pickled feeder
    (edge attribute 'geometry'); the electrical model never reads them."""
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.state = state


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("shapely"):
            try:
                return super().find_class(module, name)
            except ImportError:
                return _Geometry
        return super().find_class(module, name)


def GetDistNet(path, code):
    """extract.py:47-80: networkx graph(s) pickled as `<code>-dist-net.gpickle`
    (networkx >= 3 dropped read_gpickle; the file is a plain pickle)."""
    import networkx as nx
    load = lambda c: _Unpickler(open(f"{path}/{c}-dist-net.gpickle", "rb")).load()
    if isinstance(code, list):
        graph = nx.Graph()
        for c in code:
            graph = nx.compose(graph, load(c))
        return graph
    return load(code)


def GetCommunity(filename, com_index):
    """extract.py:82-88: line `com_index` (1-based) of space-separated home ids."""
    if not os.path.exists(f"{filename}"):
        raise ValueError(f"{filename} doesn't exist!")
    with open(f"{filename}") as f:
        lines = f.readlines()
    return [int(x) for x in lines[int(com_index) - 1].strip("\n").split(" ")]


def get_homes_ev_param(homes, dist, ev_homes, rating, capacity, initial, start, end):
    """extract.py:92-133: {residence: {"LOAD": [...], "EV": {...} or {}}}; scalar
    parameters are broadcast over ev_homes, dicts are per home."""
    as_dict = lambda v: v if isinstance(v, dict) else {h: v for h in ev_homes}
    rating, capacity, initial, start, end = map(as_dict, (rating, capacity, initial, start, end))
    evset = set(ev_homes.tolist() if hasattr(ev_homes, "tolist") else ev_homes)
    res = [n for n in dist if dist.nodes[n]["label"] == "H"]
    out = {}
    for h in res:
        out[h] = {"LOAD": [l for l in homes[h]], "EV": {}}
        if h in evset:
            out[h]["EV"] = {"rating": rating[h], "capacity": float(capacity[h]),
                            "initial": initial[h], "start": start[h], "end": end[h]}
    return out


def combine_result(P_res, P_ev, SOC, ev_homes, diff=None):
    """extract.py:135-174: the text block the reference stores under out/."""
    bar = "\n#############################################"
    sec = lambda title: f"{bar}\n{title}{bar}\n"
    fmt = lambda d, keys: "\n".join(str(h) + ":\t" + " ".join(str(y) for y in d[h]) for h in keys)
    data = sec("Residence Usage Profile") + fmt(P_res, P_res)
    data += sec("EV Charger Usage Profile") + fmt(P_ev, ev_homes)
    data += sec("EV Charger State of Charge Profile") + fmt(SOC, ev_homes)
    if diff:
        data += sec("EV Convergence over Iterations")
        data += "\n".join(str(h) + ":\t" + " ".join(str(diff[k + 1][h]) for k in range(len(diff)))
                          for h in ev_homes)
    return data
