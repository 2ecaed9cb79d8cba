import os, sys, subprocess, json
# each setting in its own process (the knobs are read once)
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
os.environ.setdefault("TQDM_DISABLE", "1")
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
n = 10_000_000
out = torch.empty(n, dtype=torch.float64, device="cuda")
for _ in range(5): m._dev.sample(n, 5, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(50): m._dev.sample(n, 5, first_id=i * n, out=out)
b.record(); torch.cuda.synchronize()
print("%%.4f ms  checksum %%d" %% (a.elapsed_time(b) / 50, int(out.view(torch.int64).sum().item())))
''' % root
for env in ({}, {"MC_SAMPLE_WG_PER_CU": "8"}, {"MC_SAMPLE_BIG_WG": "2"}, {"MC_SAMPLE_BIG_WG": "2", "MC_SAMPLE_HOT_KB": "40"},
            {"MC_SAMPLE_BIG_WG": "2", "MC_SAMPLE_HOT_KB": "60"}, {"MC_SAMPLE_BIG_WG": "4"}, {"MC_SAMPLE_BIG_WG": "4", "MC_SAMPLE_HOT_KB": "40"},
            {"MC_SAMPLE_BIG_WG": "1", "MC_SAMPLE_HOT_KB": "100"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120)
    print(env, r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
