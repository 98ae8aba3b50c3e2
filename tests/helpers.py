"""Shared helpers of the parity tests: oracle <-> engine data conversion."""
import numpy as np

from oracle import revs_oracle as ro


def oracle_homes(w):
    """revs_admm_amd.synthetic.Workload -> oracle Homes."""
    return ro.homes_from_records(w.load, w.homes)


def f32(a):
    return np.asarray(a, np.float32).astype(np.float64)
