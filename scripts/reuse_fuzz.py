#!/usr/bin/env python3
"""The list of heavy values a workspace hands from one sort to the next (include/mi355mc.h, mc_rank_build) under
fire: one workspace per size, key sets of sort_fuzz's random shapes one after the other -- often the same shape
again, often another -- through mc_rank_build and mc_order_keys; every result against torch.sort / a stable
argsort.  Usage: reuse_fuzz.py [rounds] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from montecarlo_amd import _native as nv
from montecarlo_amd.engine import DeviceRank, device_order
from sort_fuzz import make

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
ages = {}
for n in (65_536, 70_001, 300_000, 1_000_003):
    ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device="cuda").random_(0, 256)
    wso = torch.empty(nv.lib.mc_order_workspace_bytes(n), dtype=torch.uint8, device="cuda").random_(0, 256)
    srt, order = torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")
    state = None
    for it in range(rounds):
        if state is None or float(torch.rand(1, generator=g)) < 0.3:
            state = g.get_state()                           # a new source
        elif float(torch.rand(1, generator=g)) < 0.5:
            g.set_state(state)                              # the very same keys again
        x, kind = make(g, n)
        if float(torch.rand(1, generator=g)) < 0.5:
            x = x[torch.randperm(n, generator=g)]
        xd = x.cuda()
        rk = DeviceRank(xd, workspace=ws)
        info = rk.check()
        ok = torch.equal(rk.keys_sorted, torch.sort(xd)[0])
        device_order(xd, srt, order, wso)
        torch.cuda.synchronize()
        want = np.argsort(x.numpy(), kind="stable")
        ok2 = np.array_equal(order.cpu().numpy().view(np.uint32), want.astype(np.uint32)) and torch.equal(srt.cpu(), x[want])
        ages[info["list_age"]] = ages.get(info["list_age"], 0) + 1
        if not (ok and ok2):
            bad += 1
            print(f"n={n} round {it} kind {kind}: sort {'ok' if ok else 'BAD'} order {'ok' if ok2 else 'BAD'} {info}", flush=True)
    print(f"n={n}: {rounds} rounds done", flush=True)
print("list ages seen:", dict(sorted(ages.items())))
print("ALL OK" if not bad else f"{bad} BAD")
sys.exit(1 if bad else 0)
