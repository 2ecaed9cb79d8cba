#!/usr/bin/env python3
"""How much of a model's sampled keys the K commonest values hold (K = 256 .. 2^20), for the v2 and the backoff
model: the numbers behind the size of the large search's list.  Usage: key_coverage.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import torch
from montecarlo_amd import synth
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
models = {"v2": lambda: NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6),
          "backoff": lambda: BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)}
for name, make in models.items():
    m = make()
    keys = m._dev.sample(n, 5)["ml2p"]
    v, c = torch.unique(keys.view(torch.int64), return_counts=True)
    c, _ = torch.sort(c, descending=True)
    cum = torch.cumsum(c, 0).double() / n
    line = [f"{name}: n={n} distinct={v.numel()}"]
    for K in (256, 490, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1048576):
        if K <= v.numel():
            line.append(f"top{K}={cum[K-1].item():.3f} (cnt>={c[K-1].item()/n:.1e})")
    print(" ".join(line), flush=True)
    # bits: the sorted distinct values' spread
    vs = torch.sort(v).values
    print("  min/max key", keys.min().item(), keys.max().item(), flush=True)
    del keys, v, c, cum
    torch.cuda.empty_cache()
