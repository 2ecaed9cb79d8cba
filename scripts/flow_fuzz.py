#!/usr/bin/env python3
"""Differential fuzz of the estimator's flow (lib4mc/MonteCarloLib.py:9-68 as the package runs it on the GPU: rank
build, order, bisection, guess numbers, write2) against the oracle's restatement (oracle/mc_oracle.py: RankOracle)
on random inputs: 1 .. 3e5 sampled scores with random multiplicities (a few heavy values, clusters, all distinct),
0 .. 5000 scored passwords with random counts whose scores lie on sampled values, between them, below all, above all,
tie with each other or are 1022.0 (unparseable); sorted or not (need_resort), add1 or not.  Password, score, count,
cracked and ratio columns must be equal; a guess number may differ from the oracle's by the 1e-9 the positions are
allowed (sums in another order) plus one (ceil() on an exact integer, handed on by the running maximum); where all
guess numbers are equal, write2's text is compared byte for byte.  Usage: flow_fuzz.py [cases] [seed]"""
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")
from montecarlo_amd.lib4mc.MonteCarloLib import MonteCarloLib  # noqa: E402
from oracle import mc_oracle as orc  # noqa: E402

REL = 1e-9


def samples(rng, distinct_and_large=False):
    n = int(10 ** rng.uniform(0, 5.5))
    kind = int(rng.integers(0, 5))
    if distinct_and_large:                           # (the case that settles on the expanded arrays, once a run for certain)
        n, kind = 300000, 1
    if kind == 0:                                    # score-like: heavy values over a tail of repeated ones
        distinct = int(10 ** rng.uniform(0, 4)) + 1
        tail = np.round(rng.random(n) * distinct) / 64 + 3
        heavy = np.array([17.25, 21.5, 33.0078125, 40.0])
        pick = rng.integers(0, 8, size=n)
        x = np.where(pick < 4, heavy[np.minimum(pick, 3)], tail)
    elif kind == 1:                                  # all distinct (the run index refuses: the expanded arrays)
        x = rng.random(n) * 60 + 1
    elif kind == 2:                                  # few values
        x = rng.integers(0, int(rng.integers(1, 40)), size=n) / 4 + 2.0
    elif kind == 4:                                  # certain strings: -log2 p = +0.0 for many samples, tiny values beside
        x = np.where(rng.random(n) < 0.5, 0.0, rng.integers(0, 6, size=n) * 2.0 ** -int(rng.integers(1, 60)))
    else:                                            # clusters at several scales
        x = 10 + rng.integers(0, 30, size=n) + rng.integers(0, 5, size=n) * 2.0 ** -20 + (rng.random(n) < 0.1) * rng.random(n)
    return [float(v) for v in x], kind


def scored(rng, keys):
    t = int(rng.integers(0, 5001)) if rng.random() < 0.9 else 0
    k = np.asarray(keys)
    src = int(rng.integers(0, 4))
    xs = k[rng.integers(0, len(k), size=t)].copy()
    if src == 1:
        xs += rng.normal(0, 1e-3, size=t)
    elif src == 2:
        xs = rng.random(t) * (k.max() - k.min() + 4) + k.min() - 2
    elif src == 3:
        xs[rng.random(t) < 0.3] = 1022.0
        xs[rng.random(t) < 0.1] = k.min() - 1
        xs[rng.random(t) < 0.1] = k.max() + 1
    xs = np.maximum(xs, 0.0)
    if t and rng.random() < 0.3:                     # both zeros among the scores (k_score_bpe's -0.0 is the reference's)
        xs[rng.random(t) < 0.2] = -0.0
        xs[rng.random(t) < 0.1] = 0.0
    out = [(f"pw{i}", int(rng.integers(1, 50)), float(x)) for i, x in enumerate(xs)]
    resort = bool(rng.integers(0, 2))
    if not resort:
        out.sort(key=lambda r: r[2])
    return out, resort


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for c in range(cases):
        keys, kind = samples(rng, distinct_and_large=(c == 0))
        sc, resort = scored(rng, keys)
        add1 = bool(rng.integers(0, 2))
        a, b = list(keys), list(keys)
        o = orc.RankOracle(a)
        mc = MonteCarloLib(b)
        ok = a == b                                                         # both sorted the caller's list in place
        want = o.iter2gc(list(sc), need_resort=resort, add1=add1)
        got = mc.ml2p_iter2gc(list(sc), need_resort=resort, add1=add1)
        ok = ok and len(got) == len(want)
        exact = 0
        for g, w in zip(got, want):
            ok = ok and (g[0], g[1], g[2], g[4], g[5]) == (w[0], w[1], w[2], w[4], w[5])
            ok = ok and abs(int(g[3]) - int(w[3])) <= REL * int(w[3]) + 1
            exact += int(g[3]) == int(w[3])
        if os.environ.get("FLOW_FUZZ_VERBOSE") and exact < len(want):
            shown = 0
            for g, w in zip(got, want):
                if int(g[3]) != int(w[3]) and shown < 3:
                    print(f"   x={w[1]!r}: oracle estimate {o.ml2p2rank(w[1])!r} -> {w[3]}, here {float(mc.ml2p2rank(w[1]))!r} -> {g[3]}", flush=True)
                    shown += 1
        text = "-"
        if ok and exact == len(want):
            buf = io.StringIO()
            mc.write2(buf)
            same = buf.getvalue() == orc.RankOracle.format_gc(want)
            ok, text = ok and same, "ok" if same else "BAD"
        probes = [float(v) for v in rng.choice(np.asarray(keys), size=5)] + [0.0, 1e9]
        for x in probes:
            w = o.ml2p2rank(x)
            ok = ok and abs(float(mc.ml2p2rank(x)) - w) <= REL * w
        bad += 0 if ok else 1
        print(f"case {c:3d} samples={len(keys):6d} kind={kind} index={mc.device_rank.check()['index']:6s} scored={len(sc):4d} "
              f"resort={int(resort)} add1={int(add1)} equal ranks={exact}/{len(want)} write2={text} {'ok' if ok else 'BAD'}",
              flush=True)
    print("ALL OK" if not bad else f"{bad} FAILED")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
