"""Build libmi355mc.so (the C-ABI + HIP kernels) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
built .so travels to the GPU box with the tree (it is git-ignored, not
gpurun-ignored).  Usage:  python -m montecarlo_amd.build [--force] [--sanitize]
(--sanitize: the host-only sources under ASan + UBSan into libmi355mc_host_san.so, see build_sanitized)
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


HOST_SOURCES = ["mc_table.cpp", "mc_train.cpp", "mc_ingest.cpp"]       # no GPU code: table compiler, trainers, ingest, writers
HOST_SAN_LIB = os.path.join(HERE, "libmi355mc_host_san.so")
SAN_FLAGS = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=undefined", "-fPIC", "-shared", "-ffp-contract=off", "-pthread", "-Wall",
             "-Wno-unused-function"]


def sanitizer_runtime() -> str:
    """Path of the AddressSanitizer runtime g++ links against ('' when the image has none): a process that loads the
    sanitized library without being built with the sanitizer itself (python) preloads it (LD_PRELOAD)."""
    gxx = os.environ.get("CXX", "g++")
    try:
        path = subprocess.run([gxx, "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        return ""
    return os.path.realpath(path) if os.path.isabs(path) and os.path.exists(path) else ""


def build_sanitized(force: bool = False, verbose: bool = True) -> str:
    """The host-only part of the library (everything that parses caller-supplied text and dict shapes) under
    AddressSanitizer + UndefinedBehaviorSanitizer, for tests/test_sanitized_host.py.  CPU code only: GPU
    sanitizers are not available on this pool and nothing here touches the GPU."""
    deps = [os.path.join(CSRC, s) for s in HOST_SOURCES] + HEADERS
    if not force and os.path.exists(HOST_SAN_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_SAN_LIB) for d in deps):
        return HOST_SAN_LIB
    if not sanitizer_runtime():
        raise RuntimeError("no AddressSanitizer runtime in this image (g++ -print-file-name=libasan.so)")
    cmd = [os.environ.get("CXX", "g++")] + SAN_FLAGS + ["-o", HOST_SAN_LIB] + [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return HOST_SAN_LIB


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
    if "--sanitize" in sys.argv:
        print(build_sanitized(force="--force" in sys.argv))
    else:
        build(force="--force" in sys.argv)
