#!/usr/bin/env python3
"""Differential fuzz of the run-length rank index (mc_rank_runs_build / _query / _expand) against the expanded build
and torch, on sort_fuzz's random key shapes and on score-like keys (a few heavy values over a tail of repeated ones),
fresh buffers and buffers handed from case to case (the hint of one build serves the next).  The checks are those of
tests/test_gpu_runs.py (check_against_expanded); a key set the index refuses (MC_ERR_UNSUITED) is counted, not checked.
Usage: runs_fuzz.py [cases] [seed]"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (tests/test_gpu_runs.py imports its conftest by name)
from montecarlo_amd import _native as nv  # noqa: E402
from montecarlo_amd.engine import RunsBuffers  # noqa: E402


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


fz = load("sort_fuzz", os.path.join(ROOT, "scripts", "sort_fuzz.py"))
tr = load("gpu_runs", os.path.join(ROOT, "tests", "test_gpu_runs.py"))

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator().manual_seed(seed)
dev = torch.device("cuda", 0)
shared = {}
done = refused = 0
for c in range(cases):
    n = int(10 ** (1 + torch.rand(1, generator=g).item() * 5.8)) + 1
    pick = int(torch.randint(0, 3, (1,), generator=g))
    if pick == 0:
        keys, kind = fz.make(g, n)
        keys = keys.view(torch.float64)
    else:
        distinct = int(10 ** (torch.rand(1, generator=g).item() * 5)) + 1
        keys, kind = tr.score_like(g, n, distinct=distinct), f"score-like/{distinct}"
    keys = keys.contiguous().to(dev)
    keys = torch.where(torch.isfinite(keys) & (keys >= 0), keys, torch.full_like(keys, 1.5))
    if int(torch.randint(0, 8, (1,), generator=g)) == 0:     # certain strings among the samples: -log2 p = +0.0, many times
        zero = torch.rand(n, generator=g) < float(torch.rand(1, generator=g))
        keys = torch.where(zero.to(dev), torch.zeros_like(keys), keys)
        kind = f"{kind} with zeros"
    buffers = None
    if int(torch.randint(0, 2, (1,), generator=g)):
        size = 1 << max(10, (n - 1).bit_length())
        if size not in shared:
            shared[size] = RunsBuffers(size, dev)
        buffers = shared[size]
    rc, info, _ = tr.build_runs(keys, None, buffers)
    if rc == nv.ERR_UNSUITED:
        refused += 1
        continue
    assert rc == 0, (c, kind, n, rc, info)
    tr.check_against_expanded(f"case {c} {kind} n={n}", keys, None, queries=True, buffers=buffers)
    done += 1
    if c % 25 == 0:
        print(f"case {c}: {done} checked, {refused} refused", flush=True)
print(f"runs_fuzz: {cases} cases (seed {seed}): {done} checked against the expanded build, {refused} refused as unsuited: OK")
