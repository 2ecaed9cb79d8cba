"""Development aid: the rank build of config 3 alone (1.25e9 keys of the synthetic v2 6-gram model, or argv[1] keys)
for rocprofv3 kernel traces; prints the time of a build, the number of distinct values and how the keys spread over them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import _native as nv  # noqa: E402
from montecarlo_amd import synth  # noqa: E402
from montecarlo_amd.engine import DeviceRank  # noqa: E402
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_250_000_000
index = sys.argv[2] if len(sys.argv) > 2 else "sorted"
dev = torch.device("cuda", 0)
m3 = NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6)
keys = torch.empty(n, dtype=torch.float64, device=dev)
m3._dev.sample(n, 5, first_id=3 * n, out=keys)
torch.cuda.synchronize()
if index == "sorted":
    srt, pos = torch.empty_like(keys), torch.empty_like(keys)
    ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device=dev)
    mk = lambda: DeviceRank(keys, n_global=8 * n, workspace=ws, keys_sorted=srt, positions=pos)  # noqa: E731
else:
    mk = lambda: DeviceRank(keys, n_global=8 * n, index="runs")  # noqa: E731
rk = mk()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(2):
    rk = mk()
b.record()
torch.cuda.synchronize()
print(f"rank build ({index}) of {n} keys: {a.elapsed_time(b) / 2:.3f} ms; {rk.check()}", flush=True)
if index == "sorted" and os.environ.get("RANK_BIG_STATS"):
    edge = torch.ones(n, dtype=torch.bool, device=dev)
    edge[1:] = srt[1:] != srt[:-1]
    starts = edge.nonzero().flatten()
    del edge
    d = int(starts.numel())
    runs = torch.diff(starts, append=torch.tensor([n], device=dev))
    rs, _ = torch.sort(runs, descending=True)
    cum = torch.cumsum(rs, 0)
    print(f"distinct values: {d}; keys in the 1024 / 3072 / 65536 / 1e6 most frequent: "
          f"{[int(cum[min(k, d) - 1]) for k in (1024, 3072, 65536, 1000000)]}; values seen once: {int((runs == 1).sum())}, "
          f"at most 4 times: {int((runs <= 4).sum())}", flush=True)
