#!/usr/bin/env python3
"""Development aid: keys and distinct values per bin of the run index's value-range map, for blends of equal-width
and equal-depth bins, on sampled keys of the bench model.  python scripts/bin_stats.py [n]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")
from montecarlo_amd import synth  # noqa: E402
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
P, CELLS = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 4096
model = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100_000)), splitter="", threshold=10)
keys = model._dev.sample(n, 20250304)["ml2p"]
# the sample as k_rl_sniff takes it
idx = (torch.arange(16384, device=keys.device) // 16) * (n // 1024) + torch.arange(16384, device=keys.device) % 16
smp = keys[idx]
uv, inv, cnt = torch.unique(smp, return_inverse=True, return_counts=True)
heavy = uv[cnt >= 5]
tail_s = smp[cnt[inv] < 5]
lo, hi = tail_s.min().item(), tail_s.max().item()
rng = hi - lo
lo, hi = max(0.0, lo - 0.1 * rng), hi + 0.25 * rng
if len(sys.argv) > 4:
    lo, hi = float(sys.argv[3]), float(sys.argv[4])
is_heavy = torch.isin(keys, heavy)
tail = keys[~is_heavy]
print(f"n={n} heavy values {heavy.numel()} tail keys {tail.numel()} lo={lo:.3f} hi={hi:.3f}")
cell = lambda x: ((x - lo) * (CELLS / (hi - lo))).clamp(0, CELLS - 1).long()  # noqa: E731
hist = torch.bincount(cell(tail_s), minlength=CELLS).double()
before = torch.cumsum(hist, 0) - hist
lin = torch.arange(CELLS, device=keys.device).double() / CELLS
tu = torch.unique(tail)
for blend in (0.0, 0.25, 0.5, 0.75):
    f = blend * before / hist.sum() + (1 - blend) * lin
    lut = (f * P).long().clamp(max=P - 1)
    kb = torch.bincount(lut[cell(tail)], minlength=P)
    db = torch.bincount(lut[cell(tu)], minlength=P)
    q = lambda t: [int(torch.quantile(t.double(), x).item()) for x in (0.5, 0.9, 0.99, 1.0)]  # noqa: E731
    print(f"equal-depth share {blend}: keys per bin p50/p90/p99/max {q(kb)}  distinct per bin {q(db)}  "
          f"collisions est (d^2/4096, 2048 sub-ranges) max {int((db.double() ** 2 / 4096).max().item())}")
