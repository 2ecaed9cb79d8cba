"""Does a captured graph shorten the launch-bound calls?  mc_order_keys, the run-index build and the fused guess numbers
replayed from a hipGraph against the same calls issued on a stream (1e6 test passwords / 1e7 samples of the bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from montecarlo_amd import synth, _native as nv
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.engine import DeviceRank, RunsBuffers, device_order, device_query_guess_numbers

T, N = 1_000_000, 10_000_000
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
flat, dm = m._flat, m._dev
test = synth.zipf_corpus(T, oov_frac=0.01, stream=1)
syms, off = flat.encode_chars(test)
x = dm.score_ids(syms, off).contiguous()
keys = torch.empty(N, dtype=torch.float64, device="cuda")
dm.sample(N, 1, out=keys)
srt = torch.empty_like(x); order = torch.empty(T, dtype=torch.int32, device="cuda")
ws = torch.empty(nv.lib.mc_order_workspace_bytes(T), dtype=torch.uint8, device="cuda")
rb = RunsBuffers(N, keys.device)
pos = torch.empty(T, dtype=torch.float64, device="cuda"); cnt = torch.empty(T, dtype=torch.int64, device="cuda")
gr = torch.empty(T, dtype=torch.int64, device="cuda"); ge = torch.empty(1, dtype=torch.int64, device="cuda")
wsg = torch.empty(nv.lib.mc_guess_numbers_workspace_bytes(T), dtype=torch.uint8, device="cuda")
state = {}

def f_order():
    device_order(x, srt, order, ws)

def f_rank():
    state["rk"] = DeviceRank(keys, n_global=N, runs=rb)

def f_gn():
    device_query_guess_numbers(state["rk"], srt, True, out_pos=pos, out_cnt=cnt, out=gr, workspace=wsg, exact=ge)

def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

f_order(); f_rank(); f_gn(); torch.cuda.synchronize()
for name, fn in (("mc_order_keys", f_order), ("mc_rank_runs_build", f_rank), ("mc_rank_runs_guess_numbers", f_gn)):
    us_stream = timed(fn)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    us_graph = timed(g.replay)
    print(f"{name:28s} stream {us_stream:8.1f} us   graph {us_graph:8.1f} us", flush=True)
want = torch.sort(x, stable=True)
print("order still right:", bool((want[1].int() == order).all()))
