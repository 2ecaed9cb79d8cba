#!/usr/bin/env python3
"""mc_sample_carry with MC_CARRY_GROUPS copy workgroups (read once per process; PROBE_W = int64 words copied):
the sampling launch alone, with the copy into pinned host memory, with the copy into device memory, and the copy
with next to nothing sampled beside it.  Milliseconds per launch (HIP events, 1e7 samples, 8 MB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo

m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100_000)), splitter="", threshold=10)
dm = m._dev
N, W = 10_000_000, int(os.environ.get("PROBE_W", "1000002"))
keys = torch.empty(N, dtype=torch.float64, device="cuda")
src = torch.arange(W, dtype=torch.int64, device="cuda")
pin = torch.zeros(W, dtype=torch.int64).pin_memory()
dev = torch.zeros(W, dtype=torch.int64, device="cuda")


def timed(f, reps=200):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {
    "sample": timed(lambda: dm.sample(N, 1, out=keys)),
    "carry_pinned": timed(lambda: dm.sample(N, 1, out=keys, carry=(src, pin))),
    "carry_device": timed(lambda: dm.sample(N, 1, out=keys, carry=(src, dev))),
    "copy_alone_pinned": timed(lambda: dm.sample(256, 1, out=keys, carry=(src, pin))),
    "copy_alone_device": timed(lambda: dm.sample(256, 1, out=keys, carry=(src, dev))),
    "runtime_copy": timed(lambda: pin.copy_(src, non_blocking=True)),
}
assert torch.equal(pin, src.cpu()) and torch.equal(dev, src)
print("groups", os.environ.get("MC_CARRY_GROUPS", "by size"),
      {k: round(v, 4) for k, v in out.items()}, flush=True)
