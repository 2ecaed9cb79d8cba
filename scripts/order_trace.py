#!/usr/bin/env python3
"""mc_order_keys on the bench's test-set scores, a few calls, for `rocprofv3 --kernel-trace -- python3 scripts/order_trace.py`:
scripts/trace_db_breakdown.py then shows one call's kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import torch
from montecarlo_amd import synth, _native as nv
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.engine import device_order
T = 1_000_000
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
syms, off = m._flat.encode_chars(synth.zipf_corpus(T, oov_frac=0.01, stream=1))
x = m._dev.score_ids(syms, off).contiguous()
ws = torch.empty(nv.lib.mc_order_workspace_bytes(T), dtype=torch.uint8, device="cuda")
srt = torch.empty_like(x); order = torch.empty(T, dtype=torch.int32, device="cuda")
for _ in range(8):
    device_order(x, srt, order, ws)
torch.cuda.synchronize()
