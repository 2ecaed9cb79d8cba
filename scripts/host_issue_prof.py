#!/usr/bin/env python3
"""Host cost of issuing the hot path's calls (tiny inputs, so the GPU never is the limit): microseconds per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from montecarlo_amd import synth, _native as nv
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.engine import DeviceRank, device_order, device_guess_numbers

m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(20000)), splitter="", threshold=10)
dm = m._dev
n = 2000
keys = torch.empty(n, dtype=torch.float64, device="cuda")
dm.sample(n, 1, first_id=0, out=keys)
ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device="cuda")
ks, ps = torch.empty_like(keys), torch.empty_like(keys)
srt, order = torch.empty_like(keys), torch.empty(n, dtype=torch.int32, device="cuda")
wso = torch.empty(nv.lib.mc_order_workspace_bytes(n), dtype=torch.uint8, device="cuda")
rk = DeviceRank(keys, workspace=ws, keys_sorted=ks, positions=ps)
qp, qc = torch.empty_like(keys), torch.empty(n, dtype=torch.int64, device="cuda")
gr, ge = torch.empty(n, dtype=torch.int64, device="cuda"), torch.empty(1, dtype=torch.int64, device="cuda")
wsg = torch.empty(nv.lib.mc_guess_numbers_workspace_bytes(n), dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.int64).pin_memory()
side = torch.cuda.Stream()
cases = {
    "mc_sample": lambda: dm.sample(n, 1, first_id=0, out=keys),
    "mc_rank_build": lambda: DeviceRank(keys, workspace=ws, keys_sorted=ks, positions=ps),
    "mc_order_keys": lambda: device_order(keys, srt, order, wso),
    "mc_rank_query": lambda: rk.query(keys, out_pos=qp, out_cnt=qc),
    "mc_guess_numbers": lambda: device_guess_numbers(qp, qc, True, out=gr, workspace=wsg, exact=ge),
    "event(timing) record": lambda: torch.cuda.Event(enable_timing=True).record(),
    "event record": lambda: torch.cuda.Event().record(),
    "d2h copy_ pinned": lambda: pin.copy_(gr, non_blocking=True),
    "torch add (one kernel)": lambda: qp.add_(1.0),
}
for name, f in cases.items():
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:26s} host {1e6*(t1-t0)/reps:8.1f} us/call   with drain {1e6*(t2-t0)/reps:8.1f} us/call", flush=True)
