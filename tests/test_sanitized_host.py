"""The host-side C++ (table compiler, trainers, ingest, writers: everything that parses caller-supplied text and
dict shapes) under AddressSanitizer + UndefinedBehaviorSanitizer.

`python -m montecarlo_amd.build --sanitize` compiles csrc/mc_table.cpp, mc_train.cpp and mc_ingest.cpp with g++
-fsanitize=address,undefined into montecarlo_amd/libmi355mc_host_san.so; a child python preloads the sanitizer's
runtime and takes the host entry points from that library (MC_HOST_SANITIZED, montecarlo_amd/_native.py).  Run that
way: (1) tests/malformed_inputs.py -- invalid UTF-8 of every kind, NUL bytes, 1 MB lines, empty files, \\r-only line
ends, rows and contexts of a pickle-shaped import that name unknown symbols, a writer row beyond the line table;
(2) the CPU tests of the host code themselves (tables, trainers, ingest, writers, command lines) -- any report of
either sanitizer aborts the child.  CPU code only: GPU sanitizers do not exist on this pool."""
import os
import subprocess
import sys

import pytest

from montecarlo_amd import build as mcbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_malformed_inputs_plain():
    """The same inputs against the library as shipped (statuses and results, in this process)."""
    import malformed_inputs
    malformed_inputs.main()


@pytest.fixture(scope="module")
def sanitized_env():
    runtime = mcbuild.sanitizer_runtime()
    if not runtime:
        pytest.skip("no AddressSanitizer runtime in this image (g++ -print-file-name=libasan.so)")
    lib = mcbuild.build_sanitized(verbose=False)
    env = dict(os.environ)
    env.update({"LD_PRELOAD": runtime, "MC_HOST_SANITIZED": lib, "TQDM_DISABLE": "1",
                # python itself is not built for the leak checker; everything else stays fatal
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    return env


def test_malformed_inputs_under_sanitizers(sanitized_env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "malformed_inputs.py")], env=sanitized_env,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "malformed_inputs ok:" in r.stdout and " 0 entry points" not in r.stdout, r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]


def test_host_cpu_tests_under_sanitizers(sanitized_env):
    files = [os.path.join("tests", f) for f in ("test_host_tables.py", "test_oracle_golden.py", "test_cli.py")]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + files,
                       env=sanitized_env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert " passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
