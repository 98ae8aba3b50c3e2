"""Launch times of the chained iteration's middle launch and of its three parts as separate
kernels (bench-size rows: M = 2048, T = 24; a few multipliers and many violated rows per slot,
as in the binary steady state).  python tools/fused_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revs_admm_amd import _lib                      # noqa: E402
from revs_admm_amd._lib import check, ptr           # noqa: E402

lib = _lib.load()
M, T, ks, A = 2048, 24, 4, 128
rng = np.random.default_rng(0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
f64 = dict(dtype=torch.float64, device="cuda:0")
B = rng.uniform(0, 1, (M, 30))
R = up((B @ B.T + np.diag(rng.uniform(0.5, 1.0, M))) * 1e-4)
pnq = up(np.stack([rng.uniform(0, 3, (M, T)), rng.integers(20, 60, (M, T)).astype(float),
                   -rng.uniform(0, 5, (M, T))]))
y = np.zeros((M, T))
for t in range(T):
    y[rng.choice(M, 2, replace=False), t] = rng.uniform(50, 200, 2)
yd = up(y)
vs = torch.zeros(ks, M, T, **f64)
check(lib.revs_gemm_tn_f64_split(M, T, M, ptr(R), ptr(pnq), ptr(vs), ks, None))
v = vs.sum(0).cpu().numpy()
vhi, vlo = float(np.sort(v.ravel())[-2000]), -1.0          # ~80 violated rows per slot
vf, vi = torch.zeros(M, T, **f64), torch.zeros(M, T, **f64)
nblk = int(lib.revs_op_dual_blocks(M))
part = torch.zeros(nblk, T, 4, **f64)
check(lib.revs_op_dual_rows(M, T, ks, ptr(vs), ptr(pnq), ptr(yd), vlo, vhi, ptr(vf), ptr(vi), ptr(part), None, None))
cidx = torch.zeros(T, A, dtype=torch.int64, device="cuda:0")
ccnt = torch.zeros(T, dtype=torch.int32, device="cuda:0")
cval, st = torch.zeros(T, 3, A, **f64), torch.zeros(T, 8, **f64)
kfull, yhat = torch.zeros(T, A, A, **f64), torch.zeros(T, A, **f64)
info = torch.zeros(T, dtype=torch.int32, device="cuda:0")
ytr, lin = torch.zeros(M, T, **f64), torch.zeros(T, 8, **f64)
scale, eps, kadd = abs(vhi), 1e-8, 6
nfree = pnq[1].contiguous()


def sel():
    check(lib.revs_op_dual_select(M, T, ks, ptr(vs), ptr(pnq), ptr(yd), vlo, vhi, kadd, ptr(vf), ptr(vi),
                                  ptr(part), ptr(cidx), ptr(ccnt), ptr(cval), ptr(st), 1.0, None))


def model(max_pivots=300):
    check(lib.revs_op_dual_model_small(M, T, ptr(R), ptr(nfree), ptr(cidx), ptr(ccnt), ptr(cval), 5.0,
                                       1e-10, max_pivots, ptr(kfull), ptr(yhat), ptr(info), None))


def step():
    check(lib.revs_op_dual_step_pending(T, ptr(cidx), ptr(ccnt), ptr(cval), ptr(yhat), ptr(st), scale, eps,
                                        ptr(yd), M, ptr(ytr), ptr(lin), None))


def fused():
    check(lib.revs_op_dual_select_model_step(M, T, ptr(part), nblk, ptr(yd), vlo, vhi, kadd, ptr(vf), ptr(vi),
                                             ptr(cidx), ptr(ccnt), ptr(cval), ptr(st), 1.0, ptr(R), ptr(nfree),
                                             5.0, 1e-10, 300, ptr(kfull), ptr(yhat), ptr(info), scale, eps,
                                             ptr(ytr), ptr(lin), None))


def timed(label, fn, reps=200):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{label:36s} {a.elapsed_time(b) / reps * 1e3:7.2f} us per launch", flush=True)


sel()
torch.cuda.synchronize()
print("candidates per slot", ccnt.cpu().numpy().tolist())
timed("rows + selection (two kernels)", sel)
timed("small model", model)
timed("small model, one pivoting round", lambda: model(1))
timed("step (with the copy)", step)
timed("selection + model + step, one launch", fused)
print("pivots", info.cpu().numpy().tolist())
