#!/usr/bin/env python3
"""Runs mc_rank_build a few times on 1e7 samples of the headline model (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import torch

from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.engine import DeviceRank

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
keys = m._dev.sample(n, 99)["ml2p"]
m._dev.check()
rk = DeviceRank(keys)
for _ in range(10):
    DeviceRank(keys, workspace=rk._ws, keys_sorted=rk.keys_sorted, positions=rk.positions)
torch.cuda.synchronize()
print(rk.check())
