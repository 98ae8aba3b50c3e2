"""The C-ABI library builds, loads, and exports exactly what include/revs_admm.h (the boundary)
and include/revs_admm_ops.h (the operator's building blocks) declare (no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(which=("revs_admm.h", "revs_admm_ops.h")):
    names = set()
    for h in which:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(revs_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    from revs_admm_amd import _lib, build
    build.build()                     # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_header_and_binding_agree():
    from revs_admm_amd import _lib
    names = header_functions()
    assert len(names) >= 20
    assert names == sorted(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_the_boundary_header_holds_no_operator_building_block():
    """revs_admm.h is what a reference-side binding needs: the sweeps, the individual mode, the plan and its
    native loops, the communicator.  The launches those loops are made of live in revs_admm_ops.h."""
    boundary = header_functions(("revs_admm.h",))
    assert not [n for n in boundary if n.startswith(("revs_op_", "revs_gemm_", "revs_aggregate_"))]
    assert len(boundary) <= 45, len(boundary)
    assert {"revs_agent_step", "revs_plan_create", "revs_plan_newton_solve", "revs_plan_stream_run_blocks",
            "revs_plan_chain_fold_run", "revs_comm_create", "revs_residence_solve"} <= set(boundary)


def test_every_declared_symbol_is_exported(lib):
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.revs_version().decode().startswith("revs_admm_amd")


def test_home_record_layout():
    from revs_admm_amd._lib import HOME_DTYPE
    assert HOME_DTYPE.itemsize == 32
    assert [HOME_DTYPE.fields[k][1] for k in ("ev", "start", "end", "nmin", "nmax", "rating",
                                              "capacity", "initial")] == list(range(0, 32, 4))


def test_argument_validation_without_gpu(lib):
    """Bad arguments are rejected on the host, before any launch."""
    import ctypes as C
    from revs_admm_amd._lib import PDHG
    pd = PDHG()
    lib.revs_pdhg_defaults(C.byref(pd))
    assert (pd.max_iter, pd.check, pd.polish, pd.tol) == (4000, 4, 1, 0.0)
    assert lib.revs_residual_num_chunks(100000) == 25               # 4096 homes per chunk
    assert lib.revs_residual_num_chunks(10) == 1 and lib.revs_residual_num_chunks(10 ** 7) == 256
    assert lib.revs_residual_num_chunks(0) == 0
    rc = lib.revs_agent_step(0, 24, *([None] * 13), 5.0, 0, None, None)
    assert rc == -1 and b"n_homes" in lib.revs_last_error()
    rc = lib.revs_agent_step(10, 999, *([None] * 13), 5.0, 0, None, None)
    assert rc == -1 and b"T=999" in lib.revs_last_error()
    rc = lib.revs_gemm_tn_f64(4, 300, 4, None, 4, None, 300, None, 300, 0, None)
    assert rc == -1


def test_no_cpu_fallback(monkeypatch):
    """Without a GPU the product refuses to compute (and never touches the oracle)."""
    import numpy as np
    import torch
    from revs_admm_amd import _lib
    from revs_admm_amd.engine import AdmmEngine, pack_homes
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    homes = pack_homes([True, False], 4.8, 20.0, 0.2, 11, 23)
    with pytest.raises(_lib.RevsError, match="no CPU fallback"):
        AdmmEngine(np.ones(24), homes, np.ones((2, 24)), [0, 1], np.eye(2))
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librevs_admm.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.RevsError, match="missing"):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "revs_admm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_chain_accept_host_logic_matches_the_numpy_double(lib):
    """revs_newton_chain_accept is host code (no GPU): every branch against the restatement
    in tests/fake_kernels.py on random stats blocks built around the accepted pattern."""
    import ctypes as C
    import numpy as np
    from fake_kernels import FakeKernels
    fake = FakeKernels()
    rng = np.random.default_rng(0)
    T, scale, eps, A, kadd = 24, 0.04, 1e-8, 128, 6
    seen = set()
    for trial in range(400):
        s0, s1 = np.zeros((T, 8)), np.zeros((T, 8))
        s0[:, 0] = rng.choice([0.0, 1e-12, 1e-5], T) * scale          # rows: within eps or not
        s0[:, 1] = rng.normal(0, 10, T)
        s0[:, 2] = rng.integers(0, 4, T)
        s0[:, 3] = rng.integers(0, 5, T)
        s1[:, 0] = rng.choice([0.0, 1e-12], T) * scale
        s1[:, 4] = np.abs(rng.normal(0, 1e-3, T))
        s1[:, 1] = s0[:, 1] + 2e-4 * s1[:, 4]                          # Armijo passed ...
        s1[:, 2] = rng.integers(0, 6, T)
        kind = trial % 8
        if kind == 1:
            s1[rng.integers(T), 1] -= 1.0                              # ... or not
        elif kind == 2:
            s1[rng.integers(T), 0] = 1e-3 * scale                      # not converged after
        elif kind == 3:
            s0[rng.integers(T), 2] = 7                                 # more than 8 candidates
        elif kind == 4:
            s0[:, 0] = 0.0                                             # converged before
        elif kind == 5:
            s0[rng.integers(T), 2] = A + 1
        elif kind == 6:
            s0[rng.integers(T), 2] = 45                                # dense home pass expected
        few = int(rng.integers(0, 2)) if kind == 7 else 1
        outs = []
        for impl in (lib, fake):
            a, b = C.c_int32(-1), C.c_int32(-1)
            rc = impl.revs_newton_chain_accept(T, s0.ctypes.data, s1.ctypes.data, scale, eps, A, kadd,
                                               few, C.addressof(a), C.addressof(b))
            outs.append((rc, a.value, b.value) if rc else (0,))
        assert outs[0] == outs[1], (trial, outs)
        seen.add(outs[0][0])
    assert seen == {0, 1}
