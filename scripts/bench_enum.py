#!/usr/bin/env python3
"""Throughput of mc_enumerate on the bench model (one GPU): strings written per second for a few
thresholds, beside the CPU restatement of the reference's recursion at a threshold it finishes in
seconds.  A record for DESIGN.md, not the headline bench."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import torch

from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from oracle import mc_oracle as orc

m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
dev = m._dev
dev.enumerate(15.0, 4, 256)
for thr in (24.0, 28.0, 30.0, 32.0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ml2p, lens, syms = dev.enumerate(thr, 4, 256, max_nodes=1 << 33)
    dt = time.perf_counter() - t0
    print(json.dumps({"threshold": thr, "seconds_incl_d2h": dt, **dev.last_enum_info,
                      "strings_per_s": len(ml2p) / dt, "nodes_per_s": dev.last_enum_info["nodes"] / dt}), flush=True)
    del ml2p, lens, syms
probs = {ctx: {k: vals[k] for k in keys} for ctx, (vals, keys, cum) in m.nwords.items()}
t0 = time.perf_counter()
w = orc.enumerate_strings(probs, 24.0, "\x00", "\x03", 4)
dt = time.perf_counter() - t0
print(json.dumps({"cpu_restatement_threshold": 24.0, "strings": len(w), "seconds": dt, "strings_per_s": len(w) / dt}))
