"""k_sample on the v2 6-gram model alone (config 3's sampler), 1e8 samples: python scripts/exp_v2_sample.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from montecarlo_amd import synth
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
m = NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6)
dm = m._dev
keys = torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(2):
    dm.sample(n, 1, out=keys)
torch.cuda.synchronize(); dm.check()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
ev[0].record()
for i in range(5):
    dm.sample(n, 1, first_id=i * n, out=keys); ev[i + 1].record()
torch.cuda.synchronize()
cnt = dm.check()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(5))
print(f"{n} samples: median {ts[2]:.3f} ms, min {ts[0]:.3f}; draws/sample {cnt['draws'] / (5 * n):.3f}; table {m._flat.info[8] / 1e6:.1f} MB; checksum {float(keys.sum()):.6f}")
