"""Build librevs_admm.so (HIP, gfx950 only) in-tree with hipcc.

    python -m revs_admm_amd.build [--force]

The library is the product: there is no Python or CPU fallback for anything it
computes.  hipcc cross-compiles gfx950 code objects without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librevs_admm.so")
SOURCES = ["runtime.cpp", "agent_kernels.hip", "operator_kernels.hip", "newton_kernels.hip", "newton_big.hip",
           "gemm_kernels.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "tuning.h"), os.path.join(CSRC, "select_body.h"),
           os.path.join(CSRC, "tree_body.h"), os.path.join(CSRC, "internal.h"),
           os.path.join(ROOT, "include", "revs_admm.h"), os.path.join(ROOT, "include", "revs_admm_ops.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; librevs_admm.so cannot be built")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, out: str | None = None, extra=()) -> str:
    """`out` / `extra`: a tuning build beside the product's (other -D flags, another file name:
    loaded with REVS_LIB=path, see _lib.py) -- tools/ only."""
    if out is None and not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    target = out or LIB
    common = [hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC",
              "-I", os.path.join(ROOT, "include"), "-I", CSRC, *extra]
    # one hipcc per source, side by side (the operator kernels alone take over a minute), then the link
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory(prefix="revs_build_") as tmp:
        objs = [os.path.join(tmp, os.path.splitext(os.path.basename(s))[0] + ".o") for s in srcs]

        def compile_one(pair):
            src, obj = pair
            cmd = common + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            return subprocess.run(cmd, capture_output=True, text=True)

        with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
            for r in pool.map(compile_one, zip(srcs, objs)):
                if r.returncode != 0:
                    raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        link = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", target] + objs
        if verbose:
            print(" ".join(link), flush=True)
        r = subprocess.run(link, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc (link) failed:\n{r.stdout}\n{r.stderr}")
    return target


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    if args and args[0] == "--out":         # python -m revs_admm_amd.build --out lib_x.so -DREVS_...=..
        print(build(force=True, verbose=True, out=os.path.abspath(args[1]), extra=tuple(args[2:])))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
