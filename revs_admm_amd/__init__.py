"""revs_admm_amd -- MI355X-native engine for the distributed (ADMM) path of REVS.

    lpsolver       the reference's call surface (solve_ADMM, solve_residence, compute_Rmat)
    revs_fixture   REVS class (read_inputs / get_*_optimal) over the engine
    engine         array-level AdmmEngine: device buffers + kernel driver
    synthetic      synthetic feeders / residences for benchmarks and tests
    build          compiles csrc/*.hip into librevs_admm.so (gfx950)

Importing the package does not need a GPU; computing anything does.
"""
__version__ = "0.1.0"
