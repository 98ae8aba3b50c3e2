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


def tree_from_R(R, rtol=1e-12):
    """Recover a radial feeder from its LinDistFlow matrix alone (reference lpsolver.py:17-26, 184-189: callers
    of the solver hold R = 2 F D F^T restricted to the residence nodes, not the network).  For a radial feeder
    R[i][j] / 2 is the resistance shared by the substation->i and substation->j paths, R[i][i] / 2 the whole
    path to i: a tree metric.  Rows are placed by increasing depth; row i hangs below the point at depth
    c = max_j R[i][j] / 2 (over the rows placed so far) of the path to the row j that attains it -- an existing
    node where one sits at that depth, else a new junction node splitting an edge (junctions of the feeder that
    carry no residence are not rows of R).  Returns (parent, edge_r, cons_of) for AdmmEngine(feeder=...) /
    feeder_tree -- at most 2 M nodes -- or None when R is not such a matrix (asymmetric, negative, or a common
    path longer than one of the two paths).  The caller verifies the tree against R (a product with random
    injections) before trusting it."""
    R = np.asarray(R, np.float64)
    M = R.shape[0]
    if R.ndim != 2 or R.shape[1] != M or M == 0:
        return None
    scale = float(np.abs(R).max())
    if not np.isfinite(scale) or scale <= 0.0 or np.abs(R - R.T).max() > rtol * scale or R.min() < -rtol * scale:
        return None
    tol = rtol * scale
    depth = 0.5 * np.diag(R)
    order = np.argsort(depth, kind="stable")
    parent, edge_r, cons_of, ndepth = [], [], [], []       # tree nodes (rows and junctions)
    node_of_row = np.full(M, -1, np.int64)
    placed = []
    for i in order:
        di = depth[i]
        if not placed:
            c, j = 0.0, -1
        else:
            common = 0.5 * R[i, placed]
            k = int(np.argmax(common))
            c, j = float(common[k]), int(placed[k])
            if c > min(di, depth[j]) + tol:
                return None                                # a common path longer than a whole path: no tree
        if c <= tol:
            par = -1                                       # shares nothing with the others: its own lateral off the substation
            c = 0.0
        else:
            # the point at depth c on the path substation -> j: walk up from j's node
            b = int(node_of_row[j])
            a = parent[b]
            while a >= 0 and ndepth[a] >= c - tol:
                b, a = a, parent[a]
            if abs(ndepth[b] - c) <= tol:
                par = b
            else:                                          # between a and b: a junction the matrix has no row for
                par = len(parent)
                parent.append(a)
                edge_r.append(c - (ndepth[a] if a >= 0 else 0.0))
                cons_of.append(-1)
                ndepth.append(c)
                parent[b] = par
                edge_r[b] = ndepth[b] - c
        node_of_row[i] = len(parent)
        parent.append(par)
        edge_r.append(max(di - c, 0.0))
        cons_of.append(int(i))
        ndepth.append(di)
        placed.append(i)
    return np.asarray(parent, np.int64), np.asarray(edge_r, np.float64), np.asarray(cons_of, np.int64)
