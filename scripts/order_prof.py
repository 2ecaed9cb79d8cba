#!/usr/bin/env python3
"""Timing of mc_order_keys / mc_guess_numbers on the test-set shapes of bench.py and scripts/e2e_flow.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from montecarlo_amd import synth, _native as nv
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.engine import device_order, device_guess_numbers

T = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
flat, dm = m._flat, m._dev
test = synth.zipf_corpus(T, oov_frac=0.01, stream=1)
syms, off = flat.encode_chars(test)
zipf = dm.score_ids(syms, off)
ml2p_, lens_, syms_ = dm.enumerate(28.0, 4, 256)
uniq = ml2p_[:T].contiguous().double() if torch.is_tensor(ml2p_) else torch.from_numpy(np.asarray(ml2p_[:T])).cuda()
g = torch.Generator().manual_seed(1)
cases = {"zipf test set (dups)": zipf, "unique test set": uniq,
         "random": (torch.rand(T, generator=g, dtype=torch.float64) * 90).cuda(),
         "sorted already": torch.sort(uniq)[0].contiguous()}
ws = torch.empty(nv.lib.mc_order_workspace_bytes(T), dtype=torch.uint8, device="cuda")
for name, x in cases.items():
    x = x.contiguous()
    srt = torch.empty_like(x); order = torch.empty(x.numel(), dtype=torch.int32, device="cuda")
    for _ in range(3):
        device_order(x, srt, order, ws)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        device_order(x, srt, order, ws)
    b.record(); torch.cuda.synchronize()
    want = torch.sort(x, stable=True)
    ok = bool((want[1].int() == order).all()) and bool((want[0] == srt).all())
    print(f"{name:24s} t={x.numel():8d} distinct={int(torch.unique(x).numel()):8d} order {a.elapsed_time(b)/10*1e3:9.1f} us  ok={ok}", flush=True)
