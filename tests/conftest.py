import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "revs_121144.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    """Reference data files + the reference's own stored results (see
    tests/golden/make_fixtures.py)."""
    from oracle import revs_oracle as ro
    z, fd = ro.load_golden(GOLDEN)
    return z, fd


@pytest.fixture(scope="session")
def feeder_R(golden):
    """R_res of the 121144 feeder (lpsolver.py:184-189), via the tree form."""
    from oracle import revs_oracle as ro
    z, fd = golden
    R = ro.compute_Rmat_tree(fd)
    nonsub, res = fd.nonsub(), fd.res()
    pos = -np.ones(fd.n_nodes, np.int64)
    pos[nonsub] = np.arange(len(nonsub))
    ri = pos[res]
    return R[np.ix_(ri, ri)]


def golden_homes(z, tag, rating):
    from oracle import revs_oracle as ro
    res_ids = z["res_id"]
    idx = {h: i for i, h in enumerate(res_ids)}
    evi = np.array([idx[h] for h in z[tag + "_ev_homes"]])
    ev = np.zeros(len(res_ids), bool)
    ev[evi] = True
    # revs_config.yaml / revs_fixture.py:151-158 defaults: 20 kWh, 0.2, 11..23
    return ro.Homes.uniform(z["LOAD"], ev, rating, 20.0, 0.2, 11, 23), evi


@pytest.fixture(scope="session")
def gpu_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from revs_admm_amd import _lib, build
    build.build()               # no-op when librevs_admm.so is up to date (hipcc, gfx950)
    return _lib.load()          # raises if the HIP library is missing: no fallback
