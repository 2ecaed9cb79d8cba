"""Build libmi355mc.so (the C-ABI + HIP kernels) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
built .so travels to the GPU box with the tree (it is git-ignored, not
gpurun-ignored).  Usage:  python -m montecarlo_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi355mc.so")
SOURCES = ["mc_table.cpp", "mc_train.cpp", "mc_ingest.cpp", "mc_device.hip", "mc_rank.hip", "mc_sort.hip", "mc_runs.hip", "mc_gn.hip", "mc_enum.hip", "mc_comm.cpp"]
HEADERS = [os.path.join(CSRC, "mc_internal.h"), os.path.join(CSRC, "mc_sort.h"), os.path.join(os.path.dirname(HERE), "include", "mi355mc.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-ldl"]


def sources_digest() -> str:
    """sha256 over the native sources the library is built from (names and bytes): a measurement of the kernels
    (profiles/hbm_traffic_latest.json) records it, and bench.py quotes the measurement only while it still holds."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted([os.path.join(CSRC, s) for s in SOURCES] + HEADERS):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
