#!/usr/bin/env python3
"""The single T-column product of the dual Newton path (revs_gemm_tn_f64_split: R p / R^T y)
for every K-split.  (Measured this round: workgroup shapes 16x4, 16x2, 8x4, 8x8, 4x8, 4x4
waves x unroll and a contiguous row-panel layout of R all land at 10.6-12 us for M = 2048,
T = 24 -- the product is bound by launch + two memory round trips + the LDS reduction,
not by the access pattern.  A float variant of the product (float R, B rounded on the fly, as
a screening pass) took 8.7 us against 12.0 us: not worth a second code path.)
    python tools/gemm_split_bench.py [M] [T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from revs_admm_amd import _lib
from revs_admm_amd._lib import check, ptr

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
lib = _lib.load()
rng = np.random.default_rng(0)
A = torch.from_numpy(rng.normal(size=(M, M))).to("cuda:0")
B = torch.from_numpy(rng.normal(size=(M, T))).to("cuda:0")
st = torch.cuda.current_stream().cuda_stream
ref = (A.T @ B).cpu().numpy()
for ks in (1, 2, 4, 8):
    Cs = torch.zeros(ks, M, T, dtype=torch.float64, device="cuda:0")
    fn = lambda: check(lib.revs_gemm_tn_f64_split(M, T, M, ptr(A), ptr(B), ptr(Cs), ks, st))
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 300 * 1e3
    err = np.abs(Cs.sum(0).cpu().numpy() - ref).max()
    print(f"M={M} T={T} ksplit={ks}: {us:6.2f} us  "
          f"{M * M * 8 / us / 1e6:5.2f} TB/s  {2.0 * M * M * T / us / 1e6:5.1f} TFLOP/s  err {err:.1e}",
          flush=True)
