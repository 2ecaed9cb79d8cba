#!/usr/bin/env python3
"""The run-length rank build alone on sampled keys, for rocprofv3 --kernel-trace: python scripts/runs_prof.py n [reps]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")
from montecarlo_amd import _native as nv  # noqa: E402
from montecarlo_amd import synth  # noqa: E402
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "scripts"))
from runs_check import Runs, stream, vp  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cold = len(sys.argv) > 3 and sys.argv[3] == "cold"
model = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100_000)), splitter="", threshold=10)
keys = model._dev.sample(n, 20250304)["ml2p"]
r = Runs(n)
for _ in range(reps):
    if cold:
        nv.check(nv.lib.mc_rank_runs_forget(vp(r.ws), n, stream()))
    r.build(keys)
torch.cuda.synchronize()
try:
    print(r.check())
except nv.NativeError as e:
    print('check:', e)
if os.environ.get("RUNS_PHASES"):
    ph = (C.c_uint64 * 16)()
    nv.check(nv.lib.mc_rank_runs_debug_phases(1, None))
    r.build(keys)
    nv.check(nv.lib.mc_rank_runs_debug_phases(0, ph))
    ph = list(ph)
    print("phase ticks (100 MHz, summed over workgroups):", ph[:13], "longest workgroup", ph[14],
          "first start to last end", ph[15] - ((~ph[13]) & 0xFFFFFFFFFFFFFFFF) if ph[13] else None)
