#!/usr/bin/env python3
"""Throughput of the kernels behind BASELINE configs 3, 4 and the sub-word path (one GPU, synthetic
models): not the headline bench (bench.py = config 2), a record for DESIGN.md.  Each line: kernel,
workload, items/s (HIP events, median of 5 launches)."""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import numpy as np
import torch

from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.bpe_simulator import BpePcfgSim
from montecarlo_amd.nwords_simulator import NWordsMonteCarlo
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2] * 1e-3


def line(name, workload, items, secs, **kw):
    print(json.dumps({"kernel": name, "workload": workload, "items": items, "seconds": secs,
                      "items_per_s": items / secs, **kw}), flush=True)


def zipf_bpe_dir(path, rng):
    """A synthetic BPE-PCFG model directory (format of bpeX/modelreader.py:15-68) with Zipfian terminals:
    rows of 2e3..1e5 terminals, 400 structures."""
    for sub in ("grammar", "lower", "upper", "mixed_2", "mixed_3", "mixed_4", "digits", "special"):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    alpha = {"lower": "abcdefghijklmnopqrstuvwxyz", "upper": "ABCDEFGHIJKLMNOPQRSTUVWXYZ", "digits": "0123456789",
             "special": "!@#$%^&*._"}
    tag = {"lower": "L", "upper": "U", "digits": "D", "special": "S"}
    sizes = {"lower": {1: 26, 2: 600, 3: 8000, 4: 60000, 5: 100000, 6: 100000, 7: 50000, 8: 30000},
             "upper": {1: 26, 2: 300, 3: 2000}, "digits": {1: 10, 2: 100, 3: 1000, 4: 10000, 5: 30000, 6: 50000},
             "special": {1: 10, 2: 90}}
    have = []
    for sub, by_len in sizes.items():
        chars = np.array(list(alpha[sub]))
        for ln, want in by_len.items():
            cap = len(chars) ** ln
            n = min(want, cap)
            terms = set()
            while len(terms) < n:
                block = rng.choice(chars, size=(n, ln))
                terms.update("".join(r) for r in block)
            terms = sorted(terms)[:n]
            rng.shuffle(terms)
            w = 1.0 / np.arange(1, n + 1) ** 1.05
            w /= w.sum()
            with open(os.path.join(path, sub, f"{ln}.txt"), "w") as f:
                for t, p in zip(terms, w):
                    f.write(f"{t}\t{float(p)!r}\n")
            have.append((tag[sub], ln))
    structs = set()
    while len(structs) < 400:
        k = int(rng.integers(1, 4))
        segs = [have[int(i)] for i in rng.integers(0, len(have), size=k)]
        if all(a[0] != b[0] for a, b in zip(segs, segs[1:])):
            structs.add("".join(f"{t}{n}" for t, n in segs))
    structs = sorted(structs)
    w = 1.0 / np.arange(1, len(structs) + 1) ** 1.1
    w /= w.sum()
    with open(os.path.join(path, "grammar", "structures.txt"), "w") as f:
        for s_, p in zip(structs, w):
            f.write(f"{s_}\t{float(p)!r}\n")


def main():
    dev = torch.device("cuda")
    N = 10_000_000
    keys = torch.empty(N, dtype=torch.float64, device=dev)

    # config 3: nwords_v2 6-gram sampling (character-suffix contexts, word transitions)
    m = NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6)
    line("k_sample (v2, multi-character symbols)", "nwords_v2 n=6, 100k synthetic lines", N,
         timed(lambda: m._dev.sample(N, 3, out=keys)), contexts=m._flat.info[0], entries=m._flat.info[1])
    m._dev.check()

    # sub-word models: sampling and the DFS scorer
    sub = synth.subword_corpus(100000)
    m = BackWordsMonteCarlo(synth.as_file(sub), splitter="\t", threshold=10)
    line("k_sample (sub-word backoff)", "backwords --splitter tab thr 10, 100k lines", N,
         timed(lambda: m._dev.sample(N, 3, out=keys)), contexts=m._flat.info[0], entries=m._flat.info[1])
    test = list(dict.fromkeys("".join(l.split("\t")) for l in synth.subword_corpus(300000, stream=1)))
    cps, off, max_len = m._flat.encode_text(test)
    line("k_score_words (DFS over segmentations)", "same model, unique test passwords", len(test),
         timed(lambda: m._dev.score_words(cps, off, max_len, 10 ** 20)), max_len=max_len,
         note="includes H2D of the batch and workspace allocation")
    m._dev.check()

    # config 4: BPE-PCFG sampling and scoring
    with tempfile.TemporaryDirectory() as tmp:
        zipf_bpe_dir(tmp, np.random.default_rng(4))
        b = BpePcfgSim(tmp)
    line("k_sample_bpe", "synthetic BPE-PCFG: 400 structures, terminal rows up to 1e5", N,
         timed(lambda: b._dev.sample(N, 3, out=keys)), entries=b._flat.info[1])
    cnt = b._dev.check()
    res = b._dev.sample(200000, 5, want_len=True, sym_stride=b._flat.max_segments)
    pw = list(dict.fromkeys(b._flat.decode_entries(res["syms"].cpu().numpy(), res["len"].cpu().numpy())))
    cps, cls, off = b.encode(pw)
    line("k_score_bpe", "same model, unique sampled strings as test set", len(pw),
         timed(lambda: b._dev.score_bpe(cps, cls, off)), note="includes H2D of the batch",
         draws_per_sample=cnt["draws"] / (6 * N))


if __name__ == "__main__":
    main()
