"""k_sample_bpe alone on config 4's synthetic model: python scripts/exp_bpe_sample.py [n]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from montecarlo_amd import synth
from montecarlo_amd.bpe_simulator import BpePcfgSim
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 125_000_000
with tempfile.TemporaryDirectory() as d:
    synth.zipf_bpe_dir(d, np.random.default_rng(4))
    m = BpePcfgSim(d)
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
