#!/usr/bin/env python3
"""Tuning experiment (GPU box): time mc_rank_build on EXP_N (default 1e7) sampled keys."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
os.environ.setdefault("TQDM_DISABLE", "1")
import torch
from montecarlo_amd import synth, _native as nv
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.engine import DeviceRank
N = int(os.environ.get("EXP_N", "10000000"))
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
keys = m._dev.sample(N, 1)["ml2p"]
ws = torch.empty(nv.lib.mc_rank_workspace_bytes(N), dtype=torch.uint8, device="cuda")
ks, pos = torch.empty_like(keys), torch.empty_like(keys)
for _ in range(3):
    DeviceRank(keys, workspace=ws, keys_sorted=ks, positions=pos)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
ev[0].record()
for i in range(10):
    DeviceRank(keys, workspace=ws, keys_sorted=ks, positions=pos); ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))
ok = bool(torch.equal(ks.view(torch.int64), torch.sort(keys)[0].view(torch.int64)))
print(json.dumps({"ms_med": ts[5], "ms_min": ts[0], "sorted": ok}))
""" % ROOT
variants = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"]
prefetch = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"]
for v in variants:
    for pf in prefetch:
        env = dict(os.environ, MC_SORT_VARIANT=v)
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300, env=env)
        print(f"variant={v} prefetch={pf} rc={r.returncode} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-1500:]}", flush=True)
        if r.returncode != 0:
            sys.exit(r.returncode)
