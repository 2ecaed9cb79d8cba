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
        synth.zipf_bpe_dir(tmp, np.random.default_rng(4))
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
