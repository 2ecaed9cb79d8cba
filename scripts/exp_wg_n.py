"""k_sample alone: workgroups a CU against the sample count, two models: python scripts/exp_wg_n.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
which, n = sys.argv[1], int(float(sys.argv[2]))
m = (NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6) if which == "v2"
     else BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10))
dm = m._dev
keys = torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(2): dm.sample(n, 1, out=keys)
torch.cuda.synchronize(); dm.check()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
ev[0].record()
for i in range(7):
    dm.sample(n, 1, first_id=i * n, out=keys); ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(7))
print(f"{ts[3]:.4f}")
''' % ROOT
for which in ("backoff", "v2"):
    for n in ("1e6", "1e7", "3e7", "1e8"):
        row = []
        for wg in ("8", "12", "16"):
            r = subprocess.run([sys.executable, "-c", CHILD, which, n], env=dict(os.environ, MC_SAMPLE_WG_PER_CU=wg),
                               capture_output=True, text=True, timeout=300)
            row.append(f"wg {wg}: {r.stdout.strip().splitlines()[-1] if r.returncode == 0 else 'FAILED'} ms")
        print(which, n, " | ".join(row), flush=True)
