#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (it needs /root/reference, which never travels
to the GPU box):   python tests/golden/make_golden.py

The reference holds no tests or fixtures of its own, so these
files are the pin for the oracle: inputs (seeded synthetic corpora, written out
as data) and the reference's outputs on them, fp64 values as C99 hex strings.
Uniforms are injected by replacing ``random.uniform`` -- the attribute
lib4mc/ProbLib.py:41 looks up at call time -- with the Philox stream of
oracle/philox.py, one independent stream per sample id.
"""
from __future__ import annotations

import hashlib
import io
import json
import os
import random
import sys
import time

os.environ["TQDM_DISABLE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from montecarlo_amd import synth  # noqa: E402
from oracle.philox import SampleStream  # noqa: E402

from backwords.backwords_secondary_trainer import backwords_counter as ref_secondary_counter  # noqa: E402
from backwords_secondary_simulator import BackWordsSecondaryMonteCarlo  # noqa: E402
from backwords_simulator import BackWordsMonteCarlo  # noqa: E402
from bpe_simulator import BpePcfgSim  # noqa: E402
from lib4mc.MonteCarloLib import MonteCarloLib  # noqa: E402
from nwords_simulator import NWordsMonteCarlo  # noqa: E402
from nwords_simulator_v2 import NWords2MonteCarlo  # noqa: E402

SEED = 0x5EED2025


class Feed:
    def __init__(self):
        self.stream = None

    def uniform(self, a, b):
        return a + (b - a) * self.stream()


FEED = Feed()
random.uniform = FEED.uniform


def hx(x) -> str:
    return float(x).hex()


def table_rows(nwords):
    rows = []
    for ctx, (vals, keys, cum) in nwords.items():
        rows.append([ctx if isinstance(ctx, str) else list(ctx), list(keys),
                     [hx(vals[k]) for k in keys], [hx(c) for c in cum]])
    return rows


def digest(obj) -> str:
    return hashlib.sha256(json.dumps(obj, ensure_ascii=True, separators=(",", ":")).encode()).hexdigest()


def sample_block(model, n):
    items = []
    for i in range(n):
        FEED.stream = SampleStream(SEED, i)
        ml2p, pwd = model.sample1()
        items.append([hx(ml2p), pwd, FEED.stream.draws])
    return {"seed": SEED, "items": items}


def score_block(model, pwds):
    out = []
    for p in pwds:
        ml2p, parts = model.calc_ml2p(p)
        out.append([p, hx(ml2p), parts])
    return out


def write_lines(name, lines):
    with open(os.path.join(HERE, name), "w", newline="\n") as f:
        f.write("\n".join(lines) + "\n")


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, ensure_ascii=True, separators=(",", ":"))
    print(f"{name}: {os.path.getsize(os.path.join(HERE, name)) / 1024:.0f} KiB")


def model_fixture(model, table, words, test_pwds, n_samples, meta, head=40):
    rows = table_rows(table)
    return {
        "meta": meta,
        "n_ctx": len(rows),
        "n_entries": sum(len(r[1]) for r in rows),
        "table_digest": digest(rows),
        "rows_head": rows[:head],
        "words_digest": digest(sorted(words.items())),
        "scores": score_block(model, test_pwds),
        "samples": sample_block(model, n_samples),
    }


def main():
    # ---------------- corpora (data files, committed) ----------------
    train_char = synth.zipf_corpus(4000, vocab=400, seed=11)
    test_char = synth.zipf_corpus(700, vocab=400, seed=11, oov_frac=0.05, stream=1)
    test_char += ["", "a", "abc", "zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz", "ab\x03cd", "ab\x00cd", "Q" * 300,
                  test_char[0] * 25]
    train_tab = synth.subword_corpus(3000, vocab=250, seed=12)
    test_tab = ["".join(l.split("\t")) for l in synth.subword_corpus(400, vocab=250, seed=12, stream=1)]
    test_tab += ["", "xy", "notinmodel~~", test_tab[0] + test_tab[1]]
    train_v2 = synth.v2_corpus(3000, vocab=250, seed=13)
    write_lines("train_char.txt", train_char)
    write_lines("test_char.txt", test_char)
    write_lines("train_tab.txt", train_tab)
    write_lines("test_tab.txt", test_tab)
    write_lines("train_v2.txt", train_v2)
    uniq_char = list(dict.fromkeys(test_char))
    uniq_tab = list(dict.fromkeys(test_tab))

    # ---------------- fixed-order n-gram (nwords_simulator.py) ----------------
    for n in (2, 4):
        m = NWordsMonteCarlo(synth.as_file(train_char), n=n, splitter="")
        dump(f"nwords_n{n}_char.json", model_fixture(
            m, m.nwords, m.words, uniq_char, 600,
            {"class": "NWordsMonteCarlo", "train": "train_char.txt", "n": n, "splitter": ""}))
    m = NWordsMonteCarlo(synth.as_file(train_tab), n=3, splitter="\t")
    dump("nwords_n3_tab.json", model_fixture(
        m, m.nwords, m.words, uniq_tab, 400,
        {"class": "NWordsMonteCarlo", "train": "train_tab.txt", "n": 3, "splitter": "\t"}))

    # ---------------- backoff (backwords_simulator.py) ----------------
    bw10 = None
    for thr in (2, 10):
        m = BackWordsMonteCarlo(synth.as_file(train_char), splitter="", threshold=thr)
        if thr == 10:
            bw10 = m
        dump(f"backwords_thr{thr}_char.json", model_fixture(
            m, m.nwords, m.words, uniq_char, 1000,
            {"class": "BackWordsMonteCarlo", "train": "train_char.txt", "threshold": thr, "splitter": ""}))
    m = BackWordsMonteCarlo(synth.as_file(train_tab), splitter="\t", threshold=3)
    dump("backwords_thr3_tab.json", model_fixture(
        m, m.nwords, m.words, uniq_tab, 400,
        {"class": "BackWordsMonteCarlo", "train": "train_tab.txt", "threshold": 3, "splitter": "\t"}, head=6))
    m = BackWordsMonteCarlo(synth.as_file(train_tab), splitter="\t", threshold=3, max_iter=2)
    dump("backwords_thr3_tab_maxiter2.json", {
        "meta": {"class": "BackWordsMonteCarlo", "train": "train_tab.txt", "threshold": 3, "splitter": "\t",
                 "max_iter": 2},
        "scores": score_block(m, uniq_tab)})

    # ---------------- secondary: count merge + freq2prob ----------------
    cfg = {"start_chr": "\x03", "end_chr": "\x00", "max_gram": 256, "threshold": 4, "training_list": []}
    round0 = train_char[:2500]
    round1 = train_char[2500:3200]
    counts, words = ref_secondary_counter(list(round0), "", "\x03", "\x00", 0, 1, 256, 4)
    sec = {}
    m = BackWordsSecondaryMonteCarlo((counts, words, cfg))
    sec["round0"] = model_fixture(m, m.nwords, m.words, uniq_char[:200], 300, {"lines": [0, 2500]})
    counts, words = ref_secondary_counter(list(round1), "", "\x03", "\x00", 0, 1, 256, 4, nwords_dict=counts,
                                          words=words)
    m = BackWordsSecondaryMonteCarlo((counts, words, cfg))
    sec["round1"] = model_fixture(m, m.nwords, m.words, uniq_char[:200], 300, {"lines": [2500, 3200]})
    sec["meta"] = {"class": "BackWordsSecondaryMonteCarlo", "train": "train_char.txt", "threshold": 4,
                   "start_chr": "\x03", "end_chr": "\x00", "counts_digest": digest(
                       [[list(k), sorted(v.items())] for k, v in counts.items()])}
    dump("secondary_thr4_char.json", sec)

    # ---------------- v2 (sampling only; its calc_ml2p/main are defective) ----------------
    m = NWords2MonteCarlo(synth.as_file(train_v2), 4)
    tab = m._NWords2MonteCarlo__nwords
    rows = table_rows(tab)
    dump("v2_n4.json", {
        "meta": {"class": "NWords2MonteCarlo", "train": "train_v2.txt", "n": 4},
        "n_ctx": len(rows), "n_entries": sum(len(r[1]) for r in rows), "table_digest": digest(rows),
        "rows_head": rows[:40],
        "samples": sample_block(m, 600)})

    # ---------------- BPE-PCFG on a hand-made model directory ----------------
    bpe_dir = os.path.join(HERE, "bpe_model")
    make_bpe_dir(bpe_dir)
    m = BpePcfgSim(bpe_dir)
    bpe_test = ["abc123", "love12", "Abc!", "x9", "iloveyou", "pass1234", "a1b2", "12345", "ABC", "zz##", "dragon77",
                "qwerty", "aB12", "ab12cd", "ab1"]
    dump("bpe.json", {
        "meta": {"class": "BpePcfgSim", "model_dir": "bpe_model"},
        "scores": [[p, hx(m.calc_ml2p(p))] for p in bpe_test],
        "samples": sample_block(m, 500)})

    # ---------------- rank estimator (lib4mc/MonteCarloLib.py) ----------------
    ml2p = []
    for i in range(1000):
        FEED.stream = SampleStream(SEED, i)
        ml2p.append(bw10.sample1()[0])
    draw_order = list(ml2p)
    mc = MonteCarloLib(ml2p)                                   # sorts ml2p in place
    scored = bw10.parse_file(synth.as_file(test_char))
    gc = mc.ml2p_iter2gc(scored)
    buf = io.StringIO()
    mc.write2(buf)
    probes = [0.0, ml2p[0], ml2p[0] - 1e-9, ml2p[17], ml2p[-1], ml2p[-1] + 1.0, 1022.0, 33.3]
    gc0 = mc.ml2p_iter2gc(scored, add1=False)
    dump("rank.json", {
        "meta": {"model": "backwords_thr10_char.json", "test": "test_char.txt", "n": 1000, "seed": SEED},
        "draw_order": [hx(x) for x in draw_order],
        "sorted": [hx(x) for x in ml2p],
        "positions": [hx(x) for x in mc._MonteCarloLib__positions],
        "probes": [[hx(x), hx(mc.ml2p2rank(x))] for x in probes],
        "scored": [[p, c, hx(x)] for p, c, x in scored],
        "gc": [[p, hx(x), c, str(r), k, hx(ratio)] for p, x, c, r, k, ratio in gc],
        "gc_add0_ranks": [str(t[3]) for t in gc0],
        "write2": buf.getvalue()})

    # ---------------- distribution of the reference's own Mersenne-Twister sampler ----------------
    import numpy as np
    rng = random.Random(777)
    random.uniform = rng.uniform
    drawn = [bw10.sample1() for _ in range(20000)]
    random.uniform = FEED.uniform
    np.savez_compressed(os.path.join(HERE, "dist_backwords_thr10.npz"),
                        ml2p=np.array([d[0] for d in drawn]), len=np.array([len(d[1]) for d in drawn]))

    # ---------------- reference timings on the config-1 shape (reported baseline) ----------------
    if "--time" in sys.argv:
        time_reference()


def make_bpe_dir(path):
    """A small model directory in the format bpeX/modelreader.py:15-68 reads."""
    import shutil
    if os.path.isdir(path):
        shutil.rmtree(path)
    for sub in ("grammar", "lower", "upper", "mixed_2", "mixed_3", "mixed_4", "digits", "special"):
        os.makedirs(os.path.join(path, sub))

    def wr(rel, rows):
        tot = sum(w for _, w in rows)
        with open(os.path.join(path, rel), "w") as f:
            for k, w in rows:
                f.write(f"{k}\t{w / tot!r}\n")

    wr("grammar/structures.txt", [("L3D3", 30), ("L4D2", 25), ("L6", 12), ("L8", 9), ("U1L2S1", 5), ("L1D1", 2),
                                  ("D5", 7), ("U3", 2), ("L2S2", 3), ("L4D4", 10), ("DM4", 6), ("L2DM2L2", 3),
                                  ("L3L3", 4), ("TM3", 1), ("L2D1", 1)])
    wr("lower/1.txt", [("a", 5), ("x", 3), ("z", 1)])
    wr("lower/2.txt", [("ab", 4), ("bc", 3), ("zz", 2), ("cd", 1)])
    wr("lower/3.txt", [("abc", 10), ("lov", 3), ("dra", 2), ("qwe", 2), ("rty", 1)])
    wr("lower/4.txt", [("love", 9), ("pass", 8), ("word", 2)])
    wr("lower/6.txt", [("qwerty", 7), ("dragon", 5), ("abcabc", 1)])
    wr("lower/8.txt", [("iloveyou", 6), ("password", 9)])
    wr("upper/1.txt", [("A", 3), ("Q", 1)])
    wr("upper/3.txt", [("ABC", 2), ("XYZ", 1)])
    wr("digits/1.txt", [("1", 6), ("9", 2), ("2", 1)])
    wr("digits/2.txt", [("12", 8), ("77", 3), ("00", 1)])
    wr("digits/3.txt", [("123", 12), ("321", 1)])
    wr("digits/4.txt", [("1234", 10), ("2024", 2)])
    wr("digits/5.txt", [("12345", 4), ("54321", 1)])
    wr("special/1.txt", [("!", 5), ("#", 1)])
    wr("special/2.txt", [("##", 2), ("!!", 3)])
    wr("mixed_2/2.txt", [("b1", 2), ("a1", 1)])
    wr("mixed_2/4.txt", [("a1b2", 3), ("aB12", 2)])
    wr("mixed_3/3.txt", [("ab1", 1), ("a1!", 1)])


def time_reference():
    """Single-threaded reference on the BASELINE config-1 shape (100k training
    lines, 10k samples, 10k test lines) -- written to reference_timing.json."""
    train = synth.zipf_corpus(100000)
    test = synth.zipf_corpus(10000, oov_frac=0.01, stream=1)
    rng = random.Random(1)
    random.uniform = rng.uniform
    out = {"python": sys.version.split()[0], "cores_used": 1, "host_cores": os.cpu_count()}
    for name, ctor in (("nwords_n4", lambda f: NWordsMonteCarlo(f, n=4, splitter="")),
                       ("backwords_thr10", lambda f: BackWordsMonteCarlo(f, splitter="", threshold=10))):
        t0 = time.perf_counter()
        m = ctor(synth.as_file(train))
        t1 = time.perf_counter()
        ml2p = m.sample(10000)
        t2 = time.perf_counter()
        mc = MonteCarloLib(ml2p)
        t3 = time.perf_counter()
        scored = m.parse_file(synth.as_file(test))
        t4 = time.perf_counter()
        mc.ml2p_iter2gc(scored)
        t5 = time.perf_counter()
        out[name] = {"train_s": t1 - t0, "samples_per_s": 10000 / (t2 - t1), "rank_build_s": t3 - t2,
                     "scored_unique_per_s": len(scored) / (t4 - t3), "unique": len(scored),
                     "rank_query_per_s": len(scored) / (t5 - t4),
                     "n_ctx": len(m.nwords), "n_entries": sum(len(v[1]) for v in m.nwords.values())}
    random.uniform = FEED.uniform
    dump("reference_timing.json", out)


if __name__ == "__main__":
    main()
