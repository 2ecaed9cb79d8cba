#!/usr/bin/env python3
"""Differential fuzz of the key sort (through mc_rank_build) and of mc_order_keys against torch.sort: random sizes
and random mixtures of the shapes that exercise different parts of the sort (ties of every multiplicity, clusters at
several scales, single-bit keys, full bit patterns, runs, sorted / reversed input).  Usage: sort_fuzz.py [cases] [seed]
[large]: with `large`, 4e6 .. 2e7 keys a case and the rank build's large search for heavy values forced
(mc_rank_debug_search(1): the path inputs of 2^26 keys and more take by themselves); the key sort only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from montecarlo_amd.engine import DeviceRank, device_order


def make(g, n):
    """n non-negative finite doubles (as int64 bit patterns < 0x7FF0...) of a random shape"""
    kind = int(torch.randint(0, 9, (1,), generator=g))
    r = lambda hi, m=n: torch.randint(0, hi, (m,), generator=g, dtype=torch.int64)  # noqa: E731
    top = 0x7FE0000000000000
    if kind == 0:      # scores: few distinct values, Zipf-ish multiplicities
        d = max(1, int(torch.randint(1, max(2, n // 3), (1,), generator=g)))
        vals = r(1 << 62, d)
        idx = (torch.rand(n, generator=g) ** 4 * d).long().clamp(max=d - 1)
        x = vals[idx]
    elif kind == 1:    # clusters at two or three scales
        x = (r(5) << 55) + (r(9) << 30) + r(1 << int(torch.randint(1, 24, (1,), generator=g)))
    elif kind == 2:    # one bit set + small noise
        x = (torch.ones(n, dtype=torch.int64) << r(62)) + r(300)
    elif kind == 3:    # any bit pattern
        x = r(top)
    elif kind == 4:    # runs of equal keys, in order
        d = max(1, n // int(torch.randint(1, 5000, (1,), generator=g)))
        x = torch.sort(r(1 << 61, d))[0].repeat_interleave((n + d - 1) // d)[:n]
    elif kind == 5:    # ascending / descending distinct
        x = torch.arange(n, dtype=torch.int64) * int(torch.randint(1, 1 << 30, (1,), generator=g)) + (1 << 40)
        if int(torch.randint(0, 2, (1,), generator=g)):
            x = x.flip(0)
    elif kind == 6:    # one giant value + a sprinkle
        x = torch.where(torch.rand(n, generator=g) < 0.97, torch.tensor(1 << 61), r(top))
    elif kind == 7:    # order words: (small rank << 32) | index
        x = (r(max(2, n // int(torch.randint(1, 400, (1,), generator=g)))) << 32) + torch.arange(n)
    else:              # doubles of a narrow range (real scores)
        x = (torch.rand(n, generator=g, dtype=torch.float64) * 60 + 3).round(decimals=int(torch.randint(0, 9, (1,), generator=g))).view(torch.int64)
    if x.numel() != n:
        x = x[:n] if x.numel() > n else torch.cat([x, x[: n - x.numel()]])
    return x.clamp(min=0, max=top).view(torch.float64).contiguous(), kind


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    bad = 0
    if large:
        from montecarlo_amd import _native as nv
        nv.lib.mc_rank_debug_search(1)
    for c in range(cases):
        if large:
            n = (1 << 22) + int(torch.randint(0, 1 << 24, (1,), generator=g))
        else:
            n = int(10 ** (torch.rand(1, generator=g).item() * 6.5)) + 1
        x, kind = make(g, n)
        xd = x.cuda()
        want = torch.sort(xd.view(torch.int64))[0]
        rk = DeviceRank(xd, index="sorted")  # (the key sort itself: the run index has its own fuzz, scripts/runs_fuzz.py)
        info = rk.check()
        ok = torch.equal(rk.keys_sorted.view(torch.int64), want)
        ok2 = True
        if not large:
            srt, order = device_order(xd)
            wo = torch.sort(xd, stable=True)
            ok2 = torch.equal(order.long() & 0xFFFFFFFF, wo[1]) and torch.equal(srt.view(torch.int64), wo[0].view(torch.int64))
        if not (ok and ok2):
            bad += 1
        print(f"case {c:3d} kind {kind} n={n:8d} sort={'ok' if ok else 'BAD'} order={'ok' if ok2 else 'BAD'} heavy={info['heavy_values']} levels={info['levels']} tiles={info['finish_tiles']}", flush=True)
        del xd, want, rk
    print("ALL OK" if not bad else f"{bad} FAILED")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
