#!/usr/bin/env python3
"""Run-length rank index (mc_rank_runs_build) against the expanded build (mc_rank_build) and torch: values,
cumulative counts (exact) and cumulative positions (1e-9 relative) on sampled keys and on synthetic key sets,
and the time of both builds.  Usage: python scripts/runs_check.py [n_samples]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")
from montecarlo_amd import _native as nv  # noqa: E402
from montecarlo_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
stream = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)  # noqa: E731


class Runs:
    def __init__(self, n):
        self.n = n
        self.cap = nv.lib.mc_rank_runs_capacity(n)
        self.vals = torch.empty(self.cap, dtype=torch.float64, device=dev)
        self.ccum = torch.empty(self.cap, dtype=torch.int64, device=dev)
        self.wcum = torch.empty(self.cap, dtype=torch.float64, device=dev)
        self.info = torch.zeros(8, dtype=torch.int64, device=dev)
        self.ws = torch.empty(nv.lib.mc_rank_runs_workspace_bytes(n), dtype=torch.uint8, device=dev)
        nv.check(nv.lib.mc_rank_runs_forget(vp(self.ws), n, stream()))

    def build(self, keys, n_global=None):
        nv.check(nv.lib.mc_rank_runs_build(vp(keys), self.n, n_global or self.n, vp(self.vals), vp(self.ccum), vp(self.wcum),
                                           self.cap, vp(self.info), vp(self.ws), self.ws.numel(), stream()))

    def check(self):
        h = (C.c_uint64 * 8)()
        rc = nv.lib.mc_rank_runs_check(vp(self.info), stream(), h)
        return rc, list(h)


def expanded(keys, n_global=None):
    n = keys.numel()
    ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device=dev)
    nv.check(nv.lib.mc_workspace_forget(vp(ws), n, stream()))
    srt, pos = torch.empty_like(keys), torch.empty_like(keys)
    nv.check(nv.lib.mc_rank_build(vp(keys), n, n_global or n, vp(srt), vp(pos), vp(ws), ws.numel(), stream()))
    torch.cuda.synchronize()
    return srt, pos, ws


def compare(name, keys, n_global=None, expect_ok=True, queries=True):
    n = keys.numel()
    r = Runs(n)
    r.build(keys, n_global)
    rc, info = r.check()
    if not expect_ok:
        assert rc == nv.ERR_UNSUITED, (name, rc, info)
        print(f"{name}: n={n} unsuited as expected (info {info[:6]})")
        return
    assert rc == 0, (name, rc, info, nv.lib.mc_last_error())
    D = info[0]
    srt, pos, _ = expanded(keys, n_global)
    bits = srt.view(torch.int64)
    uv, cnt = torch.unique_consecutive(bits, return_counts=True)
    assert D == uv.numel(), (name, D, uv.numel())
    assert torch.equal(r.vals[:D].view(torch.int64), uv), name
    cc = torch.cumsum(cnt, 0)
    assert torch.equal(r.ccum[:D], cc), name
    want = pos[cc - 1]
    got = r.wcum[:D]
    fin = torch.isfinite(want)
    rel = ((got[fin] - want[fin]).abs() / want[fin].abs().clamp_min(1e-300)).max().item() if fin.any() else 0.0
    assert torch.equal(torch.isnan(got), torch.isnan(want)), name
    assert rel <= 1e-9, (name, rel)
    if not queries:
        print(f"{name}: n={n} D={D} index ok (no queries: a NaN among the keys makes the bisection's predicate non-monotone)")
        return
    # query and expansion
    x = torch.cat([keys[: min(n, 5000)], keys[: min(n, 1000)] + 1e-7, torch.tensor([0.0, 1e9], dtype=torch.float64, device=dev)])
    x = torch.sort(x).values if not torch.isnan(x).any() else x
    p1, c1 = torch.empty_like(x), torch.empty(x.numel(), dtype=torch.int64, device=dev)
    p2, c2 = torch.empty_like(x), torch.empty(x.numel(), dtype=torch.int64, device=dev)
    nv.check(nv.lib.mc_rank_query(vp(srt), vp(pos), n, vp(x), x.numel(), vp(p1), vp(c1), stream()))
    nv.check(nv.lib.mc_rank_runs_query(vp(r.vals), vp(r.ccum), vp(r.wcum), vp(r.info), vp(x), x.numel(), vp(p2), vp(c2), stream()))
    torch.cuda.synchronize()
    assert torch.equal(c1, c2), name
    f2 = torch.isfinite(p1)
    assert torch.equal(torch.isnan(p1), torch.isnan(p2)), name
    if f2.any():
        assert ((p1[f2] - p2[f2]).abs() <= 1e-9 * p1[f2].abs()).all(), name
    ks, ps = torch.empty_like(keys), torch.empty_like(keys)
    nv.check(nv.lib.mc_rank_runs_expand(vp(r.vals), vp(r.ccum), vp(r.wcum), vp(r.info), n, n_global or n, vp(ks), vp(ps), stream()))
    torch.cuda.synchronize()
    assert torch.equal(ks.view(torch.int64), bits), name
    fin = torch.isfinite(pos)
    assert torch.equal(torch.isnan(ps), torch.isnan(pos)), name
    if fin.any():
        assert ((ps[fin] - pos[fin]).abs() <= 1e-9 * pos[fin].abs()).all(), name
    print(f"{name}: n={n} D={D} heavy={info[3]} tail={info[4]} rel={rel:.2e} ok")


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    g = torch.Generator(device="cpu").manual_seed(7)
    # small and edge sizes on synthetic keys: few repeated values + a continuous part
    for m in (1, 2, 63, 64, 65, 4095, 4096, 4097, 20000, 70001, 300000, 1 << 20, (1 << 21) + 3):
        heavy = torch.tensor([17.25, 21.5, 33.0078125, 40.0], dtype=torch.float64)
        pick = torch.randint(0, 8, (m,), generator=g)
        tail = torch.round(torch.rand(m, generator=g, dtype=torch.float64) * 4000) / 64 + 3      # ~4000 distinct values
        keys = torch.where(pick < 4, heavy[pick.clamp(max=3)], tail).to(dev).contiguous()
        compare(f"synthetic[{m}]", keys, n_global=4 * m)
    compare("all_equal", torch.full((100000,), 12.5, dtype=torch.float64, device=dev))
    k = torch.full((100000,), 12.5, dtype=torch.float64, device=dev)
    k[5] = float("inf")
    k[99] = 0.0
    compare("with_inf_zero", k)
    k[77] = float("nan")
    k[78] = torch.tensor([-1], dtype=torch.int64).view(torch.float64).item()      # all ones: the tables' empty marker
    compare("with_nan_allones", k, queries=False)
    compare("uniform_distinct", (torch.rand(3_000_000, generator=g, dtype=torch.float64) * 60 + 5).to(dev), expect_ok=False)
    # sampled keys of the bench model
    from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
    model = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100_000)), splitter="", threshold=10)
    dm = model._dev
    for m in sorted({100_000, 1_000_000, n}):
        keys = dm.sample(m, 20250304)["ml2p"]
        torch.cuda.synchronize()
        compare(f"model[{m}]", keys)
        r = Runs(m)
        srt, pos, ws = torch.empty_like(keys), torch.empty_like(keys), torch.empty(nv.lib.mc_rank_workspace_bytes(m), dtype=torch.uint8, device=dev)
        nv.check(nv.lib.mc_workspace_forget(vp(ws), m, stream()))

        def runs_warm():
            r.build(keys)

        def runs_cold():
            nv.check(nv.lib.mc_rank_runs_forget(vp(r.ws), m, stream()))
            r.build(keys)

        def full_warm():
            nv.check(nv.lib.mc_rank_build(vp(keys), m, m, vp(srt), vp(pos), vp(ws), ws.numel(), stream()))

        def full_cold():
            nv.check(nv.lib.mc_workspace_forget(vp(ws), m, stream()))
            nv.check(nv.lib.mc_rank_build(vp(keys), m, m, vp(srt), vp(pos), vp(ws), ws.numel(), stream()))

        print(f"model[{m}] ms: runs warm {timeit(runs_warm):.4f} cold {timeit(runs_cold):.4f} | mc_rank_build warm "
              f"{timeit(full_warm):.4f} cold {timeit(full_cold):.4f}")
        rc, info = r.check()
        assert rc == 0, info
    print("runs_check done")


if __name__ == "__main__":
    main()
