#!/usr/bin/env python3
"""k_score alone (the bench's 1e6 test passwords, headline model) at several launch geometries; each setting in
its own process (MC_SCORE_WG_PER_CU / MC_SCORE_CHAINS are read once).  TESTSET=unique: 1e6 distinct strings in
random order instead.  Usage: exp_score_grid.py wg:np [wg:np ...]"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
os.environ.setdefault("TQDM_DISABLE", "1")
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
test = synth.zipf_corpus(1000000, oov_frac=0.01, stream=1)        # the bench's test set
if os.environ.get("TESTSET") == "unique":                           # 1e6 DISTINCT strings: the model's most probable ones
    ml2p_, lens_, syms_ = m._dev.enumerate(28.0, 4, 256)
    test = m._flat.decode(syms_[:1000000], lens_[:1000000])
    import random
    random.Random(4).shuffle(test)
import numpy as np
syms, off = m._flat.encode_chars(test)
d_syms = torch.from_numpy(np.ascontiguousarray(syms, dtype=np.uint32).view(np.int32)).cuda()
d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.int64)).cuda()
out = torch.empty(len(test), dtype=torch.float64, device="cuda")
for _ in range(5): m._dev.score_device(d_syms, d_off, len(test), out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(200): m._dev.score_device(d_syms, d_off, len(test), out=out)
b.record(); torch.cuda.synchronize()
print("%%.4f ms  n=%%d checksum %%d" %% (a.elapsed_time(b) / 200, len(test), int(out.view(torch.int64).sum().item())))
''' % root
for spec in sys.argv[1:]:
    wg, np_ = spec.split(":")
    e = dict(os.environ); e["MC_SCORE_WG_PER_CU"] = wg; e["MC_SCORE_CHAINS"] = np_
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=200)
    print("workgroups per CU", wg, "chains", np_, r.stdout.strip().split("\n")[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
