"""The radial feeder as a tree: host-side preparation of revs_tree_t (include/revs_admm.h) and a
numpy restatement of the kernel's three prefix sums (R p in O(nodes))."""
from __future__ import annotations

import numpy as np


def feeder_tree(parent, edge_r, cons_of, checked):
    """Host-side preparation of revs_tree_t: the feeder's nodes in DFS preorder.

    parent[i]   parent of tree node i, -1 when i hangs off the substation (a forest is fine)
    edge_r[i]   resistance of the edge from i to its parent
    cons_of[i]  constraint row (0..M-1) of node i, or -1
    checked[r]  whether row r is constrained (it carries residences, lpsolver.py:188-189)
    Returns dict(n, src, end, eo, cle, w, pack) of numpy arrays (see include/revs_admm.h; the device
    gets `pack` and `w`, the separate index arrays serve tree_voltage_host)."""
    parent = np.asarray(parent, np.int64)
    # the kernel's threads own 8 consecutive positions (16 in the shape for more than 8192 nodes):
    pad = (-len(parent)) % (8 if -(-len(parent) // 8) * 8 <= 8192 else 16)
    if pad:                                  # pad with weightless nodes hanging off the substation
        parent = np.concatenate([parent, np.full(pad, -1, np.int64)])
        edge_r = np.concatenate([np.asarray(edge_r, np.float64), np.zeros(pad)])
        cons_of = np.concatenate([np.asarray(cons_of, np.int64), np.full(pad, -1, np.int64)])
    n = len(parent)
    kids = [[] for _ in range(n)]
    roots = []
    for i in range(n):
        (roots if parent[i] < 0 else kids[parent[i]]).append(i)
    order, size = [], np.ones(n, np.int64)
    stack = [(r, False) for r in reversed(roots)]
    while stack:
        u, done = stack.pop()
        if done:
            for c in kids[u]:
                size[u] += size[c]
            continue
        order.append(u)
        stack.append((u, True))
        stack.extend((c, False) for c in reversed(kids[u]))
    if len(order) != n:
        raise ValueError("feeder: parent[] does not describe a forest")
    order = np.asarray(order, np.int64)
    pos = np.empty(n, np.int64)
    pos[order] = np.arange(n)
    end = (pos + size)[order]                                  # by preorder position
    cons = np.asarray(cons_of, np.int64)[order]
    chk = np.asarray(checked, bool)
    src = np.where((cons >= 0) & chk[np.maximum(cons, 0)], cons, -1)
    eo = np.argsort(end, kind="stable")
    cle = np.searchsorted(end[eo], np.arange(n), side="right")
    w = 2.0 * np.asarray(edge_r, np.float64)[order]
    if max(int((src + 1).max(initial=0)), int(end.max(initial=0)), int(cle.max(initial=0)), n) > 0xFFFF:
        raise ValueError("feeder: a constraint row or tree position does not fit the 16-bit fields of "
                         "revs_tree_t.pack (rows and tree nodes must be below 65535)")
    pack = ((src + 1).astype(np.uint64) | (end.astype(np.uint64) << np.uint64(16))
            | (eo.astype(np.uint64) << np.uint64(32)) | (cle.astype(np.uint64) << np.uint64(48)))
    return dict(n=n, src=src.astype(np.int32), end=end.astype(np.int32), eo=eo.astype(np.int32),
                cle=cle.astype(np.int32), w=w, pack=pack)


def tree_voltage_host(tree, p):
    """numpy restatement of the three prefix sums (tests, and the constructor's check that the
    feeder reproduces Rn): v at the checked rows, indexed like p."""
    n, src = tree["n"], tree["src"]
    inj = np.where(src >= 0, 1.0, 0.0)[:, None] * p[np.maximum(src, 0)]
    C = np.concatenate([np.zeros((1, p.shape[1])), np.cumsum(inj, 0)])
    wp = tree["w"][:, None] * (C[tree["end"]] - C[:-1])
    pre = np.cumsum(wp, 0)
    F = np.concatenate([np.zeros((1, p.shape[1])), np.cumsum(wp[tree["eo"]], 0)])
    v = pre - F[tree["cle"]]
    out = np.zeros_like(p)
    out[src[src >= 0]] = v[src >= 0]
    return out
