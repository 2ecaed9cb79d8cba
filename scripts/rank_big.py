#!/usr/bin/env python3
"""mc_rank_build at config 3's per-GPU share (1.25e9 keys, 10 GB per array): time and the sort's counters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import torch
from montecarlo_amd import synth, _native as nv
from montecarlo_amd.engine import DeviceRank
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_250_000_000
models = {"v2 6-gram (config 3)": lambda: NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6),
          "backoff thr 10 (config 2)": lambda: BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)}
for name, make in models.items():
    m = make()
    keys = m._dev.sample(n, 5)["ml2p"]
    m._dev.check()
    ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    ks, ps = torch.empty_like(keys), torch.empty_like(keys)
    rk = DeviceRank(keys, n_global=8 * n, workspace=ws, keys_sorted=ks, positions=ps)
    info = rk.check()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        DeviceRank(keys, n_global=8 * n, workspace=ws, keys_sorted=ks, positions=ps)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f"{name}: n={n} rank build {ms:.2f} ms = {n/ms/1e6:.2f}e9 keys/s, {40*n/ms/1e9:.2f} TB/s of 40 B/key; workspace {ws.numel()/2**30:.1f} GiB; {info}", flush=True)
    del keys, ws, ks, ps, rk, m
    torch.cuda.empty_cache()
