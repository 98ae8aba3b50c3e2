#!/usr/bin/env python3
"""Micro-benchmark of the operator's f64 matrix-core products (one MI355X).
    python tools/gemm_bench.py [M] [T]"""
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
dev = "cuda:0"
rng = np.random.default_rng(0)
A = [torch.from_numpy(rng.normal(size=(M, M))).to(dev) for _ in range(4)]
B = [torch.from_numpy(rng.normal(size=(M, T))).to(dev) for _ in range(2)]
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for ks in (1, 2, 4):
    C0 = torch.zeros(ks, M, T, dtype=torch.float64, device=dev)
    C1 = torch.zeros_like(C0)
    # alternate between two matrix pairs, as the operator does (V,U then VT,UT)
    k = [0]

    def pair():
        i = k[0] & 1
        k[0] += 1
        check(lib.revs_gemm_tn_f64_x2(M, T, M, ptr(A[2 * i]), ptr(B[0]), ptr(C0), ptr(A[2 * i + 1]),
                                      ptr(B[1]), ptr(C1), ks, st))
    us = timeit(pair)
    byt = 2 * M * M * 8
    print(f"M={M} T={T} pair ksplit={ks}: {us:7.2f} us  {byt / us / 1e6:6.2f} TB/s matrix stream  "
          f"{2 * 2.0 * M * M * T / us / 1e6:6.2f} TFLOP/s")
for ks in (1, 2, 4, 8):
    C0 = torch.zeros(ks, M, T, dtype=torch.float64, device=dev)
    C1 = torch.zeros_like(C0)
    if 2 * T > 192:
        break
    k = [0]

    def cat():
        i = k[0] & 1
        k[0] += 1
        check(lib.revs_gemm_tn_f64_cat(M, T, M, ptr(A[i]), ptr(B[0]), ptr(B[1]), ptr(C0), ptr(C1),
                                       ks, st))
    us = timeit(cat)
    print(f"M={M} T={T} cat  ksplit={ks}: {us:7.2f} us  {M * M * 8 / us / 1e6:6.2f} TB/s matrix stream  "
          f"{2 * 2.0 * M * M * T / us / 1e6:6.2f} TFLOP/s")
C0 = torch.zeros(4, M, T, dtype=torch.float64, device=dev)
C1 = torch.zeros_like(C0)
ref = (A[0].T @ B[0]).cpu().numpy()
check(lib.revs_gemm_tn_f64_x2(M, T, M, ptr(A[0]), ptr(B[0]), ptr(C0), ptr(A[1]), ptr(B[1]), ptr(C1), 4, st))
print("max err", np.abs(C0.sum(0).cpu().numpy() - ref).max())
