"""CPU oracle for the REVS ADMM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this module.  The product (`revs_admm_amd/`) never does; it fails loudly
when its HIP library is missing instead of falling back to anything here.

This is a float64 numpy restatement of the algorithm in the reference's
`lpsolver.py` (rounak-meyur/revs-admm).  The reference builds one Gurobi model
per residence and per operator step; Gurobi (gurobipy) is a third-party,
licence-gated dependency that is not vendored in the reference and is not
installed here, so the reference cannot be executed.  Each model is small and
fully specified by lpsolver.py, so this file restates the *optimisation
problems* and solves them exactly:

  home problem      lpsolver.py:44-160   -> home_solve_binary / home_solve_relaxed
  operator problem  lpsolver.py:163-238  -> utility_solve (+ utility_kkt)
  R matrix          lpsolver.py:17-26    -> compute_Rmat / compute_Rmat_tree
  ADMM loop         lpsolver.py:242-290  -> solve_ADMM
  individual mode   lpsolver.py:407-460  -> solve_residence

Parity pin (tests/test_oracle.py): the reference's own stored results
under out/121144-com2/ (tests/golden/revs_121144.npz).  `diff[1]` of all 267 EV
homes of the stored distributed run is reproduced to 1e-12, the stored
individual-mode schedules are optimal for the restated model with identical
objective value, and the stored final schedules are feasible for it.  Later
`diff[k]` cannot be reproduced bit-for-bit by ANY re-implementation: the home
MIQP has many exactly tied optima (flat tariff blocks x repeated load values),
Gurobi's branch-and-bound picks one of them by internal order, and every later
iterate depends on that pick.  DESIGN.md states which quantities are
tie-invariant and therefore pinned.

Arrays, not dicts: homes are rows.  N homes, T slots, M constraint nodes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# feasibility slack used when turning the SOC window into slot counts; Gurobi's
# default FeasibilityTol is 1e-6, the quantities compared are O(1).
_SOC_TOL = 1e-9
SOC_TARGET = 0.9     # lpsolver.py:109   s[T] >= 0.9
SOC_MAX = 1.0        # lpsolver.py:102-103  ub = 1.0


# --------------------------------------------------------------------------
# feeder / R matrix
# --------------------------------------------------------------------------
@dataclass
class Feeder:
    """Radial network as arrays, nodes and edges in the reference graph's order."""
    label: np.ndarray      # (n_nodes,) 'S','T','R','H'
    edge_u: np.ndarray     # (n_edges,) node index
    edge_v: np.ndarray
    edge_r: np.ndarray     # resistance
    node_id: np.ndarray | None = None

    @property
    def n_nodes(self):
        return len(self.label)

    def nonsub(self):
        """Indices of the non-substation nodes, graph order (lpsolver.py:20-21)."""
        return np.where(self.label != b"S")[0]

    def res(self):
        """Indices of residences 'H', graph order (lpsolver.py:167)."""
        return np.where(self.label == b"H")[0]


def compute_Rmat(fd: Feeder) -> np.ndarray:
    """R = 2 F D F^T over the non-substation nodes -- lpsolver.py:17-26, literally.

    A is the oriented node-edge incidence matrix (networkx convention: -1 at the
    edge's first node, +1 at its second), F = inv(A[nonsub,:].T), D = diag(r).
    """
    n, e = fd.n_nodes, len(fd.edge_u)
    A = np.zeros((n, e))
    A[fd.edge_u, np.arange(e)] = -1.0
    A[fd.edge_v, np.arange(e)] = 1.0
    F = np.linalg.inv(A[fd.nonsub(), :].T)
    return 2.0 * (F * fd.edge_r[None, :]) @ F.T


def compute_Rmat_tree(fd: Feeder) -> np.ndarray:
    """Same matrix from the tree: R[i,j] = 2 * sum of r over the edges common to
    the root->i and root->j paths.  Used to cross-check compute_Rmat and for
    feeders too large for the dense inverse."""
    n = fd.n_nodes
    adj = [[] for _ in range(n)]
    for k, (u, v) in enumerate(zip(fd.edge_u, fd.edge_v)):
        adj[u].append((v, k))
        adj[v].append((u, k))
    root = int(np.where(fd.label == b"S")[0][0])
    nonsub = fd.nonsub()
    pos = -np.ones(n, dtype=np.int64)
    pos[nonsub] = np.arange(len(nonsub))
    R = np.zeros((len(nonsub), len(nonsub)))
    # DFS order; row of a child = row of its parent, plus 2r on the child's subtree
    parent = -np.ones(n, dtype=np.int64)
    pedge = -np.ones(n, dtype=np.int64)
    order, stack, seen = [], [root], np.zeros(n, bool)
    seen[root] = True
    while stack:
        u = stack.pop()
        order.append(u)
        for v, k in adj[u]:
            if not seen[v]:
                seen[v] = True
                parent[v], pedge[v] = u, k
                stack.append(v)
    # subtree membership via reverse order
    sub = [None] * n
    for u in reversed(order):
        s = [u]
        for v, _ in adj[u]:
            if parent[v] == u:
                s.extend(sub[v])
        sub[u] = s
    for u in order:
        if u == root:
            continue
        idx = pos[np.array(sub[u])]
        R[np.ix_(idx, idx)] += 2.0 * fd.edge_r[pedge[u]]
    return R


# --------------------------------------------------------------------------
# homes
# --------------------------------------------------------------------------
@dataclass
class Homes:
    """Per-residence data (extract.py:92-133 get_homes_ev_param, as arrays)."""
    LOAD: np.ndarray                 # (N,T) kW
    ev: np.ndarray                   # (N,) bool
    rating: np.ndarray               # (N,) kW      "rating"
    capacity: np.ndarray             # (N,) kWh     "capacity"
    initial: np.ndarray              # (N,)         "initial"
    start: np.ndarray                # (N,) int     "start"
    end: np.ndarray                  # (N,) int     "end"

    @property
    def N(self):
        return self.LOAD.shape[0]

    @property
    def T(self):
        return self.LOAD.shape[1]

    @staticmethod
    def uniform(LOAD, ev, rating, capacity, initial, start, end):
        N = LOAD.shape[0]
        f = lambda v, dt: np.full(N, v, dtype=dt)
        return Homes(np.asarray(LOAD, float), np.asarray(ev, bool), f(rating, float),
                     f(capacity, float), f(initial, float), f(start, np.int64),
                     f(end, np.int64))

    def window(self):
        """(N,T) bool: slots where charging is allowed (lpsolver.py:97-98)."""
        t = np.arange(self.T)[None, :]
        return (t >= self.start[:, None]) & (t < self.end[:, None]) & self.ev[:, None]


def slot_count_bounds(h: Homes):
    """Number of full-rate slots allowed by the SOC rows (lpsolver.py:101-109):
    s_T = init + n*rate/cap must lie in [0.9, 1.0] (and never below init)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        per = np.where(h.ev, h.rating / h.capacity, 1.0)
    nmin = np.ceil((np.maximum(SOC_TARGET, h.initial) - h.initial) / per - _SOC_TOL)
    nmax = np.floor((SOC_MAX - h.initial) / per + _SOC_TOL)
    nmin = np.maximum(nmin, 0).astype(np.int64)
    nmax = nmax.astype(np.int64)
    return np.where(h.ev, nmin, 0), np.where(h.ev, nmax, 0)


def _soc(h: Homes, p):
    """s_0 = init, s_t = s_{t-1} + p_{t-1}/cap (lpsolver.py:105-108); all zero
    for a residence without EV (lpsolver.py:76-79)."""
    s = np.zeros((h.N, h.T + 1))
    s[:, 0] = h.initial
    s[:, 1:] = h.initial[:, None] + np.cumsum(p, axis=1) / h.capacity[:, None]
    return np.where(h.ev[:, None], s, 0.0)


def home_linear_term(cost, h: Homes, p_est, p_sch, gamma, kappa):
    """Coefficient q_t of p_t after expanding lpsolver.py:112-128 with g = p + LOAD:
        obj = sum_t (kappa/2) g_t^2 + (c_t - a_t) g_t,
        a_t = gamma_t + (kappa/2)(p_est_t + p_sch_t)             (lpsolver.py:118-119)
      => in p:  (kappa/2) p_t^2 + q_t p_t + const,  q_t = kappa*LOAD_t + c_t - a_t."""
    a = gamma + 0.5 * kappa * (p_est + p_sch)
    return kappa * h.LOAD + np.asarray(cost)[None, :] - a


def home_objective(cost, h: Homes, p, p_est, p_sch, gamma, kappa):
    """Value of the reference objective lpsolver.py:112-128 at schedule p."""
    g = p + h.LOAD
    a = gamma + 0.5 * kappa * (p_est + p_sch)
    return ((np.asarray(cost)[None, :] - a) * g + 0.5 * kappa * g * g).sum(axis=1)


def home_solve_binary(cost, h: Homes, p_est, p_sch, gamma, kappa=5.0, tie=None):
    """Exact optimum of the reference's per-residence MIQP (lpsolver.py:44-129).

    p_t = e_t * rating with e_t binary and zero outside [start,end); the SOC rows
    reduce to n_min <= sum e_t <= n_max because p >= 0 makes s nondecreasing.
    The objective is separable, so switching slot t on costs
        delta_t = rating * ( (kappa/2)(rating + 2 LOAD_t) + c_t - a_t )
    and the optimum takes the n_min cheapest slots, then further slots while
    delta_t < 0, up to n_max.  Ties are broken towards the EARLIER slot (Gurobi
    breaks them by its own internal order; see module docstring).

    Returns p (N,T), s (N,T+1), g (N,T), status (N,) [0 ok, 1 infeasible --
    the reference prints 'No solution found' and exits, lpsolver.py:153-155].
    """
    q = home_linear_term(cost, h, p_est, p_sch, gamma, kappa)
    delta = h.rating[:, None] * (0.5 * kappa * h.rating[:, None] + q)
    win = h.window()
    nmin, nmax = slot_count_bounds(h)
    d = np.where(win, delta, np.inf)
    if tie is None:
        order = np.argsort(d, axis=1, kind="stable")      # ties -> earlier slot
    else:
        # another rule among EXACTLY tied slots (Gurobi's is unknown): a numpy Generator
        # draws the order, "last" prefers the later slot.  Used by the tests to measure how
        # much of a trajectory statistic is decided by the tie rule alone.
        sec = -np.arange(h.T)[None, :].repeat(h.N, 0) if tie == "last" else tie.random(d.shape)
        order = np.lexsort((sec, d), axis=1)
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.arange(h.T)[None, :].repeat(h.N, 0), axis=1)
    take = win & ((rank < nmin[:, None]) | ((rank < nmax[:, None]) & (delta < 0)))
    nwin = win.sum(axis=1)
    status = (h.ev & ((nmin > nmax) | (nmin > nwin))).astype(np.int32)
    p = np.where(take, h.rating[:, None], 0.0)
    p[status == 1] = 0.0
    return p, _soc(h, p), p + h.LOAD, status


def home_solve_binary_bruteforce(cost, h: Homes, p_est, p_sch, gamma, kappa=5.0):
    """Enumerate every on/off pattern (T <= ~16) and keep the feasible minimum of
    the literal objective and SOC rows.  Validates home_solve_binary's reduction."""
    T = h.T
    assert T <= 16
    pats = ((np.arange(1 << T)[:, None] >> np.arange(T)[None, :]) & 1).astype(float)
    best = np.full(h.N, np.inf)
    for i in range(h.N):
        if not h.ev[i]:
            best[i] = home_objective(cost, _row(h, i), np.zeros((1, T)), p_est[i:i+1],
                                     p_sch[i:i+1], gamma[i:i+1], kappa)[0]
            continue
        t = np.arange(T)
        ok = (pats[:, (t < h.start[i]) | (t >= h.end[i])] == 0).all(axis=1)
        P = pats * h.rating[i]
        s = h.initial[i] + np.cumsum(P, axis=1) / h.capacity[i]
        ok &= (s <= SOC_MAX + 1e-9).all(axis=1) & (s >= h.initial[i] - 1e-9).all(axis=1)
        ok &= s[:, -1] >= SOC_TARGET - 1e-9
        if not ok.any():
            continue
        g = P[ok] + h.LOAD[i][None, :]
        a = gamma[i] + 0.5 * kappa * (p_est[i] + p_sch[i])
        best[i] = ((np.asarray(cost) - a)[None, :] * g + 0.5 * kappa * g * g).sum(1).min()
    return best


def _row(h: Homes, i):
    return Homes(h.LOAD[i:i+1], h.ev[i:i+1], h.rating[i:i+1], h.capacity[i:i+1],
                 h.initial[i:i+1], h.start[i:i+1], h.end[i:i+1])


def energy_bounds(h: Homes):
    """[E_lo, E_hi] on sum_t p_t implied by init <= s_t <= 1, s_T >= 0.9 when p >= 0."""
    lo = (np.maximum(SOC_TARGET, h.initial) - h.initial) * h.capacity
    hi = (SOC_MAX - h.initial) * h.capacity
    return np.where(h.ev, lo, 0.0), np.where(h.ev, hi, 0.0)


def home_solve_relaxed(cost, h: Homes, p_est, p_sch, gamma, kappa=5.0):
    """Exact optimum of the CONTINUOUS relaxation of lpsolver.py:44-129
    (0 <= p_t <= rating in the window instead of p_t in {0, rating}) -- the
    'T-slot box+SOC-constrained QP' BASELINE.json's north_star names.

    KKT: p_t = clip(u_t + nu, 0, ub_t), u_t = -q_t/kappa, one multiplier nu for the
    only SOC rows that can bind (the terminal ones).  nu is found exactly: bisection
    to identify the linear piece, then the closed form on that piece.
    """
    q = home_linear_term(cost, h, p_est, p_sch, gamma, kappa)
    u = -q / kappa
    ub = np.where(h.window(), h.rating[:, None], 0.0)
    Elo, Ehi = energy_bounds(h)
    f = lambda nu: np.clip(u + nu[:, None], 0.0, ub).sum(axis=1)
    s0 = f(np.zeros(h.N))
    tgt = np.where(s0 < Elo, Elo, np.where(s0 > Ehi, Ehi, np.nan))
    need = ~np.isnan(tgt)
    cap = ub.sum(axis=1)
    status = (h.ev & (Elo > cap + 1e-12)).astype(np.int32)
    lo = np.full(h.N, -1.0)
    hi = np.full(h.N, 1.0)
    span = np.abs(u).max() + ub.max() + 1.0
    lo *= span
    hi *= span
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        below = f(mid) < tgt
        lo = np.where(below, mid, lo)
        hi = np.where(below, hi, mid)
    nu = 0.5 * (lo + hi)
    # closed form on the identified piece
    x = u + nu[:, None]
    free = (x > 0) & (x < ub)
    sat = (x >= ub) & (ub > 0)
    nfree = free.sum(axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        nu_exact = (tgt - (ub * sat).sum(axis=1) - (u * free).sum(axis=1)) / nfree
    nu = np.where(need & (nfree > 0), nu_exact, nu)
    nu = np.where(need, nu, 0.0)
    p = np.clip(u + nu[:, None], 0.0, ub)
    p[status == 1] = 0.0
    return p, _soc(h, p), p + h.LOAD, status


def pdhg_steps(h: Homes, tau_scale=0.25, sigma_scale=4.0):
    """Step sizes of the PDHG restatement below and of the HIP kernel
    (revs_admm_amd/csrc/agent_kernels.hip: pdhg_step_sizes).  The kernel works in
    the normalised variable x_t = p_t/rating (objective x^2/2 + b x), where the SOC
    operator is K = delta * (inclusive prefix sum), delta = rating/capacity, with
    ||K|| <= delta * T_w * 2/pi * (1 + 1/T_w) (T_w = window length)."""
    Tw = np.maximum(h.end - h.start, 1).astype(float)
    delta = np.where(h.ev, h.rating / h.capacity, 1.0)
    nK = delta * (2.0 / math.pi) * (Tw + 1.0)
    return tau_scale / nK, sigma_scale / nK, delta


def home_solve_relaxed_pdhg(cost, h: Homes, p_est, p_sch, gamma, kappa=5.0,
                            iters=2000, tol=1e-7, check=8,
                            tau_scale=None, sigma_scale=None, dtype=np.float64, full_rows=True):
    """PDHG (Chambolle-Pock) on the relaxed home QP, restating the HIP kernel's
    iteration so the two can be compared step for step.  x = p/rating in [0,w];
        x+ = clip((x - tau (K^T y + b)) / (1 + tau), 0, w)
        y+ = v - sigma clip(v / sigma, lo, hi),  v = y + sigma K (2 x+ - x)
    rows j = 0..T-1 of K x are s_{j+1} - init; lo_j = 0 (j < T-1), lo_{T-1} =
    max(0.9, init) - init, hi_j = 1 - init.  Stops when the largest change of x
    over a block of `check` iterations' last step falls below tol."""
    dt = dtype
    q = home_linear_term(cost, h, p_est, p_sch, gamma, kappa)
    rate = np.where(h.ev, h.rating, 1.0)
    b = (q / (kappa * rate[:, None])).astype(dt)
    w = h.window().astype(dt)
    if not full_rows:
        return _pdhg_presolved(h, b, w, iters, tol, check, tau_scale or 0.5, sigma_scale or 2.0, dt)
    tau, sig, delta = pdhg_steps(h, tau_scale or 0.25, sigma_scale or 4.0)
    tau, sig, delta = (tau.astype(dt)[:, None], sig.astype(dt)[:, None],
                       delta.astype(dt)[:, None])
    lo = np.zeros((h.N, h.T), dt)
    lo[:, -1] = np.maximum(SOC_TARGET, h.initial) - h.initial
    hi = np.repeat((SOC_MAX - h.initial)[:, None], h.T, 1).astype(dt)
    x = np.zeros((h.N, h.T), dt)
    y = np.zeros((h.N, h.T), dt)
    done = ~h.ev.copy()
    n_it = np.zeros(h.N, np.int64)
    one = dt(1.0)
    for k in range(iters):
        act = ~done
        if not act.any():
            break
        KTy = delta * np.cumsum(y[:, ::-1], axis=1, dtype=dt)[:, ::-1]
        xn = np.clip((x - tau * (KTy + b)) / (one + tau), 0, w)
        xb = xn + (xn - x)
        v = y + sig * (delta * np.cumsum(xb, axis=1, dtype=dt))
        yn = v - sig * np.clip(v / sig, lo, hi)
        dx = np.abs(xn - x).max(axis=1)
        dy = (np.abs(yn - y) / sig).max(axis=1)
        x = np.where(act[:, None], xn, x)
        y = np.where(act[:, None], yn, y)
        n_it += act
        if (k + 1) % check == 0:
            done |= np.maximum(dx, dy) <= tol
    p = x.astype(float) * np.where(h.ev, h.rating, 0.0)[:, None]
    return p, _soc(h, p), p + h.LOAD, n_it


def _pdhg_presolved(h: Homes, b, w, iters, tol, check, tau_scale, sigma_scale, dt):
    """PDHG on the presolved home QP (the kernel's default): with p >= 0 the SOC is
    nondecreasing, so of init <= s_t <= 1, s_T >= 0.9 only the terminal rows can bind;
    K is the single row delta*1^T (||K|| = delta sqrt(T_w)) and the dual one scalar."""
    Tw = np.maximum(np.minimum(h.end, h.T) - np.maximum(h.start, 0), 1).astype(dt)[:, None]
    delta = np.where(h.ev, h.rating / h.capacity, 1.0).astype(dt)[:, None]
    nK = delta * np.sqrt(Tw)
    tau, sig = dt(tau_scale) / nK, dt(sigma_scale) / nK
    lo = (np.maximum(SOC_TARGET, h.initial) - h.initial).astype(dt)[:, None]
    hi = (SOC_MAX - h.initial).astype(dt)[:, None]
    x = np.zeros(b.shape, dt)
    y = np.zeros((h.N, 1), dt)
    done = ~h.ev.copy()
    n_it = np.zeros(h.N, np.int64)
    one = dt(1.0)
    for k in range(iters):
        act = ~done
        if not act.any():
            break
        xn = np.clip((x - tau * (sig * delta * y + b)) / (one + tau), 0, w)
        v = y + delta * (xn + (xn - x)).sum(axis=1, keepdims=True, dtype=dt)
        yn = v - np.clip(v, lo, hi)
        dx = np.abs(xn - x).max(axis=1)
        dy = np.abs(yn - y)[:, 0]
        x = np.where(act[:, None], xn, x)
        y = np.where(act[:, None], yn, y)
        n_it += act
        if (k + 1) % check == 0:
            done |= np.maximum(dx, dy) <= tol
    p = x.astype(float) * np.where(h.ev, h.rating, 0.0)[:, None]
    return p, _soc(h, p), p + h.LOAD, n_it


# --------------------------------------------------------------------------
# operator ("Utility") problem
# --------------------------------------------------------------------------
def utility_g0(p_est, p_sch, gamma, kappa):
    """Unconstrained minimiser of lpsolver.py:196-207:
        obj = sum_i (kappa/2)|g_i|^2 + g_i . a_i,  a_i = gamma_i - (kappa/2)(p_est_i + p_sch_i)
      => g0 = -a/kappa = (p_est + p_sch)/2 - gamma/kappa."""
    return 0.5 * (p_est + p_sch) - gamma / kappa


def voltage_limits(vset, vlow, vhigh):
    """lpsolver.py:185-186: limits on R g are v^2 - vset^2."""
    return vlow * vlow - vset * vset, vhigh * vhigh - vset * vset


def utility_solve(Rn, node_of, g0, kappa, vlo, vhi, eps=1e-11, max_iter=20000,
                  rho_b=None, rho_v=None, alpha=1.6, sigma=1e-6, check=25,
                  warm=None, return_info=False):
    """Operator step, lpsolver.py:163-238:  min (kappa/2)|g - g0|^2  s.t.  g >= 0
    (Gurobi's default variable lower bound, lpsolver.py:179-180) and
    vlo <= Rn (A g) <= vhi per slot (lpsolver.py:191-193), A = home->node
    aggregation (identity for the reference feeder, where every residence is its
    own node: Rn = R_res, lpsolver.py:188-189).

    Solved by ADMM in OSQP form (x, z=Cx, y) on C = [Rn A; I] with the KKT matrix
    inverted once -- plain dense linear algebra in home space, i.e. deliberately
    NOT the node-space Woodbury form the HIP path uses.  Runs to eps ~1e-11 so it
    can serve as ground truth; `utility_kkt` certifies any candidate solution."""
    N, T = g0.shape
    M = Rn.shape[0]
    A = np.zeros((M, N))
    A[node_of, np.arange(N)] = 1.0
    RA = Rn @ A                                            # (M,N)
    nr2 = np.linalg.norm(Rn, 2) ** 2 * max(1.0, np.bincount(node_of, minlength=M).max())
    rho_b = kappa if rho_b is None else rho_b
    rho_v = 25.0 * kappa / nr2 if rho_v is None else rho_v
    K = (kappa + sigma + rho_b) * np.eye(N) + rho_v * (RA.T @ RA)
    Ki = np.linalg.inv(K)
    q = -kappa * g0
    if warm is None:
        x = np.maximum(g0, 0.0)
        zv, zb = np.clip(RA @ x, vlo, vhi), x.copy()
        yv, yb = np.zeros((M, T)), np.zeros((N, T))
    else:
        x, zv, zb, yv, yb = (w.copy() for w in warm)
    it = 0
    for it in range(1, max_iter + 1):
        xt = Ki @ (sigma * x - q + RA.T @ (rho_v * zv - yv) + (rho_b * zb - yb))
        zvt = RA @ xt
        x = alpha * xt + (1 - alpha) * x
        hv = alpha * zvt + (1 - alpha) * zv
        hb = alpha * xt + (1 - alpha) * zb
        zv_n = np.clip(hv + yv / rho_v, vlo, vhi)
        zb_n = np.maximum(hb + yb / rho_b, 0.0)
        yv = yv + rho_v * (hv - zv_n)
        yb = yb + rho_b * (hb - zb_n)
        zv, zb = zv_n, zb_n
        if it % check == 0:
            rp = max(np.abs(RA @ x - zv).max() / max(abs(vhi), abs(vlo), 1e-30),
                     np.abs(x - zb).max())
            rd = np.abs(kappa * (x - g0) + RA.T @ yv + yb).max() / kappa
            if max(rp, rd) <= eps:
                break
    g = zb                                                  # exactly >= 0
    if return_info:
        return g, dict(iters=it, state=(x, zv, zb, yv, yb), yv=yv, yb=yb)
    return g


def utility_kkt(Rn, node_of, g, g0, kappa, vlo, vhi, yv, yb):
    """KKT certificate of lpsolver.py:163-238 for a candidate g with multipliers
    (yv of the voltage rows, (M,T); yb of the g >= 0 rows, (N,T), as returned by
    utility_solve's info).  Returns (primal infeasibility, stationarity residual
    relative to kappa, sign/complementarity violation).  Multipliers of a
    degenerate vertex are not unique, so they are an input, not recomputed."""
    N, T = g.shape
    M = Rn.shape[0]
    A = np.zeros((M, N))
    A[node_of, np.arange(N)] = 1.0
    RA = Rn @ A
    v = RA @ g
    prim = max((v - vhi).max(), (vlo - v).max(), (-g).max(), 0.0)
    stat = np.abs(kappa * (g - g0) + RA.T @ yv + yb).max() / kappa
    scale = max(abs(vhi), abs(vlo))
    # yv > 0 only on rows at vhi, yv < 0 only on rows at vlo, yb <= 0 only where g = 0
    comp = max(np.abs(np.maximum(yv, 0) * (vhi - v)).max() / scale,
               np.abs(np.minimum(yv, 0) * (v - vlo)).max() / scale,
               np.abs(yb * g).max(), np.maximum(yb, 0).max())
    return prim, stat, comp / kappa


def utility_solve_dual(Rn, node_of, g0, kappa, vlo, vhi, y0=None, max_iter=80, tol=1e-10, kadd=8,
                       nonneg=True):
    """The same operator step (lpsolver.py:163-238) through its dual, slot by slot --
    about a second where `utility_solve`'s home-space ADMM takes tens of seconds on the
    121144 feeder, so that golden-feeder trajectories and runs of hundreds of ADMM
    iterations stay affordable in tests.  The QP is strictly convex, so ANY point carrying
    a KKT certificate is THE solution: every answer of this routine is certified against the
    literal problem by `utility_kkt` before it is returned (RuntimeError otherwise -- the
    caller then falls back to `utility_solve`), and tests/test_oracle.py compares the two
    solvers directly.

    Per slot, with B = Rn A (rows: constraint nodes, columns: residences) and y the
    multipliers of the voltage rows (y > 0: at vhi, y < 0: at vlo):
        g(y) = max(g0 - B^T y / kappa, 0)        (stationarity with the g >= 0 rows)
    Iteration: working set W = rows with y != 0 plus the `kadd` most violated rows; with
    the free residences F = {g0 - B^T y / kappa > 0} frozen, the dual restricted to W is
        max_u>=0  -1/2 u^T K u + c^T u,   K = S B_WF B_WF^T S / kappa,  c = S (B_WF g0_F - b_W)
    (S = row side +-1, b = vhi / vlo), solved exactly as a non-negative least-squares
    problem on K's Cholesky factor (duplicated rows make K singular; the sign constraint
    is what keeps the multipliers of near-identical rows from cancelling).  Repeat until no
    row is violated and every row with a multiplier sits on its bound.
    Returns g (N,T), yv (M,T), yb (N,T) -- multipliers in `utility_kkt`'s convention.
    `nonneg=False` drops the g >= 0 rows: NOT the reference's model, used only as a negative
    control by the tests (Gurobi's default lb = 0, lpsolver.py:179-180, is easy to miss)."""
    from scipy.linalg import cholesky, solve_triangular
    from scipy.optimize import nnls
    N, T = g0.shape
    M = Rn.shape[0]
    B = Rn[:, node_of]                                        # (M,N)
    Y = np.zeros((M, T)) if y0 is None else np.array(y0, float)
    G = np.zeros((N, T))
    scale = max(abs(vlo), abs(vhi), 1e-300)
    for t in range(T):
        y, g0t = Y[:, t].copy(), g0[:, t]
        ok = False
        for _ in range(max_iter):
            d = B.T @ y / kappa
            free = (g0t - d > 0) if nonneg else np.ones(N, bool)
            g = np.where(free, g0t - d, 0.0)
            v = B @ g
            viol = np.where(y == 0, np.maximum(np.maximum(v - vhi, vlo - v), 0.0), 0.0)
            act = np.where(y > 0, np.abs(v - vhi), np.where(y < 0, np.abs(v - vlo), 0.0))
            if viol.max() <= tol * scale and act.max() <= tol * scale:
                ok = True
                break
            cand = np.argsort(-viol, kind="stable")[:kadd]
            W = np.union1d(np.where(y != 0)[0], cand[viol[cand] > tol * scale])
            up = np.where(y[W] != 0, y[W] > 0, v[W] > vhi)
            sgn, b = np.where(up, 1.0, -1.0), np.where(up, vhi, vlo)
            BW = B[np.ix_(W, np.where(free)[0])]
            K = sgn[:, None] * (BW @ BW.T / kappa) * sgn[None, :]
            c = sgn * (BW @ g0t[free] - b)
            K[np.diag_indices_from(K)] += 1e-12 * np.trace(K) / len(W)
            L = cholesky(K, lower=True)
            u, _ = nnls(L.T, solve_triangular(L, c, lower=True), maxiter=50 * len(W))
            y = np.zeros(M)
            y[W] = sgn * u
        if not ok:
            raise RuntimeError(f"utility_solve_dual: slot {t} did not settle")
        Y[:, t] = y
        G[:, t] = g
    if not nonneg:          # (negative control of the tests: no certificate for a wrong model)
        return G, Y, None
    yb = -(kappa * (G - g0) + B.T @ Y)                        # multipliers of the g >= 0 rows
    yb[G > 0] = 0.0
    prim, stat, comp = utility_kkt(Rn, node_of, G, g0, kappa, vlo, vhi, Y, yb)
    gs = max(1.0, np.abs(g0).max())
    # (complementarity is |y| x distance-to-bound: relative to the largest multiplier)
    if not (prim <= 1e-8 * scale and stat <= 1e-9 * gs
            and comp <= 1e-8 * max(gs, np.abs(Y).max() / kappa)):
        raise RuntimeError(f"utility_solve_dual: KKT certificate failed {prim:.2e} {stat:.2e} {comp:.2e}")
    return G, Y, yb


# --------------------------------------------------------------------------
# ADMM loop
# --------------------------------------------------------------------------
@dataclass
class ADMMTrace:
    diff: np.ndarray                 # (iters, N)  lpsolver.py:284
    P_est: list = field(default_factory=list)
    P_sch: list = field(default_factory=list)
    G: list = field(default_factory=list)
    S: list = field(default_factory=list)
    util_iters: list = field(default_factory=list)


def solve_ADMM(h: Homes, Rn, node_of, cost, kappa=5.0, iter_max=15, vset=1.0,
               vlow=0.95, vhigh=1.05, mode="binary", keep=False, util_eps=1e-11,
               home_solver=None, util_method="admm", variant=None):
    """lpsolver.py:242-290.  State starts at zero (lines 244-246).  Each iteration:
      1. operator step from (P_est[k], P_sch[k], G[k])            (lines 256-259)
      2. every home from (P_est[k], P_sch[k], G[k]) -- NOTE the OLD estimate
         P_est[k], not the one just computed                       (line 273)
      3. G[k+1] = G[k] + (kappa/2)(P_est[k+1] - P_sch[k+1]),
         diff[k+1] = |P_est[k+1] - P_sch[k+1]|_2 / T                (lines 280-284)
    Returns (diff (iters,N), P_sch, S=p_ev, C=soc) like the reference's
    (diff, P_sch[k], S[k], C[k]); with keep=True also the full trace."""
    N, T = h.LOAD.shape
    vlo, vhi = voltage_limits(vset, vlow, vhigh)
    P_est = np.zeros((N, T))
    P_sch = np.zeros((N, T))
    G = np.zeros((N, T))
    solver = home_solver or (home_solve_binary if mode == "binary" else home_solve_relaxed)
    diffs = np.zeros((iter_max, N))
    tr = ADMMTrace(diffs)
    warm = None
    S = C = None
    y_warm = None
    for k in range(iter_max):
        g0 = utility_g0(P_est, P_sch, G, kappa)
        info = None
        if variant == "no_lb":
            # NEGATIVE CONTROL (tests only): the operator QP without Gurobi's default lb = 0
            # (lpsolver.py:179-180) -- a plausible misreading of the reference
            P_est_new, y_warm, _ = utility_solve_dual(Rn, node_of, g0, kappa, vlo, vhi, y0=y_warm,
                                                      nonneg=False)
        elif util_method == "dual":
            try:
                P_est_new, y_warm, _ = utility_solve_dual(Rn, node_of, g0, kappa, vlo, vhi, y0=y_warm)
                info = dict(iters=0)
            except RuntimeError:
                y_warm = None
        if info is None and variant != "no_lb":
            P_est_new, info = utility_solve(Rn, node_of, g0, kappa, vlo, vhi, eps=util_eps,
                                            warm=warm, return_info=True)
            warm = info["state"]
        info = info or dict(iters=0)
        # NEGATIVE CONTROL (tests only): homes solved from the NEW estimate P_est[k+1]
        # instead of P_est[k] (contradicts lpsolver.py:273)
        pe_for_homes = P_est_new if variant == "new_estimate" else P_est
        S, C, P_sch_new, status = solver(cost, h, pe_for_homes, P_sch, G, kappa)
        if status is not None and np.ndim(status) and status.dtype == np.int32 and status.any():
            raise RuntimeError("No solution found (lpsolver.py:153-155)")
        check = P_est_new - P_sch_new
        G = G + 0.5 * kappa * check
        diffs[k] = np.linalg.norm(check, axis=1) / T
        P_est, P_sch = P_est_new, P_sch_new
        if keep:
            tr.P_est.append(P_est.copy()); tr.P_sch.append(P_sch.copy())
            tr.G.append(G.copy()); tr.S.append(S.copy())
            tr.util_iters.append(info["iters"])
    if keep:
        return diffs, P_sch, S, C, tr
    return diffs, P_sch, S, C


# --------------------------------------------------------------------------
# centralized problem the distributed scheme decomposes
# --------------------------------------------------------------------------
def solve_central_lp(tariff, h: Homes, Rn, node_of, vset, vlow, vhigh, binary=False,
                     time_limit=120.0):
    """The network-wide problem whose consensus form the reference's ADMM iterates on:
        min  sum_h tariff . g_h
        s.t. every residence's rows of Home (lpsolver.py:83-109: charger window and rating,
             SOC recursion, init <= s_t <= 1, s_T >= 0.9)   -- the residences' side
             g >= 0, vlo <= Rn (A g)[:, t] <= vhi per slot (lpsolver.py:179-193) -- the
             operator's side                                 (consensus: the same g).
    At a fixed point of lpsolver.py:254-287 the residence minimises (c - G).g over its rows
    and the operator G.g over the network rows with the same g: the KKT system of this LP.
    It is the yard-stick of the reference's test-centralopt.py, which compares a stored
    distributed schedule with `Central(homes, dist, COST, ...)` per residence,
    dev = 100 (C2 - C1) / C1 (test-centralopt.py:114-116; the `Central` class lives in the
    reference's un-vendored libs/pySchedEVChargelib).  NOTE lpsolver.solve_central
    (lpsolver.py:463-502) is a different model (no s_T >= 0.9 row, sign-flipped voltage
    rows): its optimum is "no charging" and it is not what the ADMM converges to.

    binary=False: chargers relaxed to 0 <= p <= rating (an LP, HiGHS through scipy);
    binary=True: p = e * rating, e binary (a MILP; small cases only).
    Returns p (N,T), g (N,T), per-residence cost (N,), total cost."""
    from scipy import sparse
    from scipy.optimize import Bounds, LinearConstraint, milp
    N, T = h.LOAD.shape
    M = Rn.shape[0]
    vlo, vhi = voltage_limits(vset, vlow, vhigh)
    c = np.asarray(tariff, float)
    win = h.window()
    Elo, Ehi = energy_bounds(h)
    # variables x[i,t] = p[i,t] / rating_i in [0, win]
    rate = np.where(h.ev, h.rating, 0.0)
    cost = (rate[:, None] * c[None, :]).ravel()
    ub = win.astype(float).ravel()
    cons = []
    # SOC: with p >= 0 the SOC is nondecreasing, so init <= s_t <= 1 and s_T >= 0.9 reduce to
    # Elo <= sum_t p_t <= Ehi (slot_count_bounds / energy_bounds)
    rows = np.repeat(np.arange(N), T)
    Asoc = sparse.csr_matrix((np.repeat(rate, T), (rows, np.arange(N * T))), shape=(N, N * T))
    cons.append(LinearConstraint(Asoc, Elo, Ehi))
    # voltage rows per slot: Rn[:, node_of] (LOAD + p)[:, t] in [vlo, vhi]
    Bm = Rn[:, node_of]                                       # (M,N)
    base = Bm @ h.LOAD                                        # (M,T)
    Bs = sparse.csr_matrix(Bm * rate[None, :])
    sel = [sparse.csr_matrix((np.ones(N), (np.arange(N), np.arange(N) * T + t)), shape=(N, N * T))
           for t in range(T)]
    Av = sparse.vstack([Bs @ sel[t] for t in range(T)]).tocsr()
    cons.append(LinearConstraint(Av, (vlo - base).T.ravel(), (vhi - base).T.ravel()))
    integrality = (np.ones(N * T) if binary else np.zeros(N * T))
    r = milp(cost, constraints=cons, bounds=Bounds(0.0, ub), integrality=integrality,
             options=dict(time_limit=time_limit, presolve=True))
    if r.x is None:
        raise RuntimeError(f"solve_central_lp: {r.message}")
    p = r.x.reshape(N, T) * rate[:, None]
    g = p + h.LOAD
    if g.min() < -1e-9:
        raise RuntimeError("solve_central_lp: g >= 0 violated by the base load")
    per_home = g @ c
    return p, g, per_home, float(per_home.sum())


def solve_central_ref(tariff, h: Homes, R_res, vset, vmin, vmax, time_limit=120.0):
    """lpsolver.solve_central itself (lpsolver.py:463-502) as a MILP through HiGHS:
        min  sum_h tariff . g_h,   g_h = p_h + LOAD_h           (add_home_load, objective_centralized)
        p_h = e_h rating_h, e binary, e = 0 outside [start, end); init <= s_t <= 1 with
        s_t = init + sum_{tau < t} p_tau / capacity -- NO s_T >= 0.9 row   (add_home_EV, 338-379)
        vmin^2 - vset^2 <= -R_res g[:, t] <= vmax^2 - vset^2             (network_constraints, 386-405)
    one residence per row of R_res (the reference's `res` order).  Returns p (N,T), g (N,T),
    total cost; raises RuntimeError where the reference prints 'No solution found'."""
    from scipy import sparse
    from scipy.optimize import Bounds, LinearConstraint, milp
    N, T = h.LOAD.shape
    c = np.asarray(tariff, float)
    vlo, vhi = vmin * vmin - vset * vset, vmax * vmax - vset * vset
    win = h.window()
    rate = np.where(h.ev, h.rating, 0.0)
    cost = (rate[:, None] * c[None, :]).ravel()              # in e; the LOAD term is a constant
    cons = []
    # p >= 0: the SOC is nondecreasing, so init <= s_t <= 1 is sum_t p_t / capacity <= 1 - init
    rows = np.repeat(np.arange(N), T)
    Asoc = sparse.csr_matrix((np.repeat(rate / np.where(h.ev, h.capacity, 1.0), T), (rows, np.arange(N * T))),
                             shape=(N, N * T))
    cons.append(LinearConstraint(Asoc, -np.inf, np.where(h.ev, 1.0 - h.initial, 0.0) + 1e-12))
    base = -(R_res @ h.LOAD)                                  # (N,T): -R_res LOAD
    Bs = sparse.csr_matrix(-R_res * rate[None, :])
    sel = [sparse.csr_matrix((np.ones(N), (np.arange(N), np.arange(N) * T + t)), shape=(N, N * T))
           for t in range(T)]
    Av = sparse.vstack([Bs @ sel[t] for t in range(T)]).tocsr()
    cons.append(LinearConstraint(Av, (vlo - base).T.ravel(), (vhi - base).T.ravel()))
    r = milp(cost, constraints=cons, bounds=Bounds(0.0, win.astype(float).ravel()),
             integrality=np.ones(N * T), options=dict(time_limit=time_limit, presolve=True))
    if r.x is None:
        raise RuntimeError(f"solve_central_ref: {r.message}")
    p = np.round(r.x).reshape(N, T) * rate[:, None]
    g = p + h.LOAD
    return p, g, float((g @ c).sum())


# --------------------------------------------------------------------------
# individual mode
# --------------------------------------------------------------------------
def solve_residence(tariff, h: Homes):
    """lpsolver.py:407-460: min 0.01 * tariff.g + 0.99 * (1 - s_T) with the binary
    charger and the SOC box but NO s_T >= 0.9 row (add_home_EV, lines 338-379).
    Switching slot t on changes the objective by 0.01 c_t rate - 0.99 rate/cap, so
    the optimum takes the cheapest slots while that is negative, up to n_max.
    Ties -> earlier slot.  Returns p, s, g."""
    c = np.asarray(tariff)[None, :]
    win = h.window()
    _, nmax = slot_count_bounds(h)
    delta = 0.01 * c * h.rating[:, None] - 0.99 * (h.rating / h.capacity)[:, None]
    d = np.where(win, delta, np.inf)
    order = np.argsort(d, axis=1, kind="stable")
    rank = np.empty_like(order)
    np.put_along_axis(rank, order, np.arange(h.T)[None, :].repeat(h.N, 0), axis=1)
    take = win & (rank < nmax[:, None]) & (delta < 0)
    p = np.where(take, h.rating[:, None], 0.0)
    return p, _soc(h, p), p + h.LOAD


def residence_objective(tariff, h: Homes, p):
    g = p + h.LOAD
    sT = np.where(h.ev, h.initial + p.sum(1) / h.capacity, 0.0)
    return 0.01 * (g * np.asarray(tariff)[None, :]).sum(1) + 0.99 * (1.0 - sT)


# --------------------------------------------------------------------------
# synthetic workloads (shared by tests and bench so both sides see one input)
# --------------------------------------------------------------------------
def homes_from_records(load, rec) -> Homes:
    """float64 view of the per-residence records the device gets (revs_home_t)."""
    return Homes(np.asarray(load, float), rec["ev"].astype(bool), rec["rating"].astype(float),
                 rec["capacity"].astype(float), rec["initial"].astype(float),
                 rec["start"].astype(np.int64), rec["end"].astype(np.int64))


def load_golden(path):
    z = np.load(path)
    fd = Feeder(z["node_label"], z["edge_u"], z["edge_v"], z["edge_r"], z["node_id"])
    return z, fd
