#!/usr/bin/env python3
"""Quick check of mc_rank_build's sort on one GPU against torch.sort, on the key shapes the tests use, with the
sort's own counters (heavy values, tail keys, levels, finish tiles).  A development aid, not a test."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import numpy as np
import torch

from montecarlo_amd import _native as nv
from montecarlo_amd import synth
from montecarlo_amd.engine import DeviceRank


def check(name, keys, reps=0):
    keys = keys.contiguous().cuda()
    rk = DeviceRank(keys, index="sorted")
    info = rk.check()
    srt, _ = torch.sort(keys)
    ok = torch.equal(rk.keys_sorted.view(torch.int64), srt.view(torch.int64))
    line = f"{name:28s} n={keys.numel():>11d} ok={ok} info={info}"
    if not ok:
        bad = (rk.keys_sorted.view(torch.int64) != srt.view(torch.int64)).nonzero().flatten()
        line += f" first_bad={int(bad[0])} n_bad={bad.numel()}"
    if reps:
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            DeviceRank(keys, workspace=rk._ws, keys_sorted=rk.keys_sorted, positions=rk.positions)
        b.record()
        torch.cuda.synchronize()
        line += f" ms={a.elapsed_time(b) / reps:.4f}"
    print(line, flush=True)
    return ok


def main():
    g = torch.Generator(device="cpu").manual_seed(3)
    ok = True
    for n in (1, 2, 63, 64, 65, 4095, 4096, 4097, 6143, 6144, 6145, 12289, 65535, 65536, 100001):
        keys = torch.rand(n, generator=g, dtype=torch.float64) * 60
        if n > 10:
            keys[n // 2:] = keys[: n - n // 2].clone()
        ok &= check(f"uniform+dups", keys)
    ok &= check("all equal", torch.full((5000,), 12.5, dtype=torch.float64))
    ok &= check("all equal big", torch.full((300000,), 12.5, dtype=torch.float64))
    n = 1_000_003
    up = torch.arange(n, dtype=torch.float64) * 1e-4 + 3.0
    g = torch.Generator(device="cpu").manual_seed(11)
    cases = {
        "ascending": up,
        "descending": up.flip(0),
        "two values": torch.where(torch.rand(n, generator=g) < 0.999, 17.25, 17.5).double(),
        "one heavy bin": torch.where(torch.rand(n, generator=g) < 0.98, torch.tensor(40.0, dtype=torch.float64),
                                     torch.rand(n, generator=g, dtype=torch.float64) * 200),
        "all bit patterns": torch.randint(0, 0x7FEFFFFFFFFFFFFF, (n,), generator=g, dtype=torch.int64)
        .view(torch.float64),
        "runs": torch.repeat_interleave(torch.rand(n // 1000 + 1, generator=g, dtype=torch.float64) * 90, 1000)[:n],
        "clusters": (torch.randint(0, 3, (n,), generator=g).double() * 16 + 5
                     + torch.randint(0, 1 << 20, (n,), generator=g).double() * 2.0 ** -45),
        "nested clusters": ((torch.randint(0, 5, (n,), generator=g) << 50) + (torch.randint(0, 7, (n,), generator=g) << 25)
                            + torch.randint(0, 1 << 12, (n,), generator=g) + (1 << 60)).view(torch.float64),
        "order words": ((torch.randint(0, 300, (n,), generator=g) ** 2 << 32) + torch.arange(n) + (1 << 52))
        .view(torch.float64),
        "geometric": ((torch.ones(n, dtype=torch.int64) << torch.randint(8, 61, (n,), generator=g))
                      + torch.randint(0, 200, (n,), generator=g)).view(torch.float64),
    }
    for name, keys in cases.items():
        ok &= check(name, keys, reps=5)
    if len(sys.argv) > 1 and sys.argv[1] == "model":
        from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
        m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
        for n in (200_000, 3_000_000, 10_000_000):
            keys = m._dev.sample(n, 99)["ml2p"]
            m._dev.check()
            ok &= check("backoff model samples", keys, reps=20)
    print("ALL OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
