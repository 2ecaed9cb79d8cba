#!/usr/bin/env python3
"""k_sample alone (N=1e7 samples of MODEL=backoff|v2|ngram|backoff3; backoff is the headline model) at several launch geometries; each setting in its own process
(the knobs MC_SAMPLE_WG_PER_CU / MC_SAMPLE_HOT_KB / MC_SAMPLE_CHAINS are read once).  Usage: exp_sample_grid.py 8 12 16 ..."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
os.environ.setdefault("TQDM_DISABLE", "1")
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
which = os.environ.get("MODEL", "backoff")
n = int(float(os.environ.get("N", "1e7")))
if which == "v2":
    from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
    m = NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6)
elif which == "ngram":
    from montecarlo_amd.nwords_simulator import NWordsMonteCarlo
    m = NWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), n=4, splitter="")
elif which == "backoff3":
    m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(30000, stream=3)), splitter="", threshold=3)
else:
    m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
out = torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(5): m._dev.sample(n, 5, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(100): m._dev.sample(n, 5, first_id=i * n, out=out)
b.record(); torch.cuda.synchronize()
print("%%.4f ms  checksum %%d" %% (a.elapsed_time(b) / 100, int(out.view(torch.int64).sum().item())))
''' % root
for spec in sys.argv[1:]:
    wg, _, skew = spec.partition(":")
    e = dict(os.environ); e["MC_SAMPLE_WG_PER_CU"] = wg
    if skew:
        e["MC_SAMPLE_SKEW"] = skew
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120)
    print("workgroups per CU", spec, r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
