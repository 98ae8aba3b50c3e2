"""Shared helpers of the parity tests: oracle <-> engine data conversion."""
import numpy as np

from oracle import revs_oracle as ro


def oracle_homes(w):
    """revs_admm_amd.synthetic.Workload -> oracle Homes (float64 view of the same
    records the device gets)."""
    h = w.homes
    return ro.Homes(np.asarray(w.load, float), h["ev"].astype(bool), h["rating"].astype(float),
                    h["capacity"].astype(float), h["initial"].astype(float),
                    h["start"].astype(np.int64), h["end"].astype(np.int64))


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)
