"""Seeded synthetic Zipfian password corpora (SURVEY.md §8(d) "Synthetic inputs").

There is no network and the reference ships no data, so every benchmark and test
input is produced here.  The law: a vocabulary of ``vocab`` base strings, lengths
uniform in [4, 12] over a 41-character alphabet, drawn with Zipf(s) frequencies.
Test sets come from the same law plus a fraction of out-of-model strings.
"""
from __future__ import annotations

import io
from typing import List

import numpy as np

ALPHABET = "abcdefghijklmnopqrstuvwxyz0123456789!@#._"
assert len(ALPHABET) == 41

DEFAULT_SEED = 20250304


def _vocabulary(rng: np.random.Generator, vocab: int) -> List[str]:
    lens = rng.integers(4, 13, size=vocab)
    chars = np.array(list(ALPHABET))
    # characters themselves are skewed so that n-gram rows have uneven mass
    w = 1.0 / np.arange(1, len(ALPHABET) + 1) ** 0.7
    w /= w.sum()
    out = []
    seen = set()
    for ln in lens:
        while True:
            s = "".join(rng.choice(chars, size=int(ln), p=w))
            if s not in seen:
                seen.add(s)
                out.append(s)
                break
    return out


def zipf_corpus(lines: int, vocab: int = 20000, s: float = 1.1, seed: int = DEFAULT_SEED,
                oov_frac: float = 0.0, stream: int = 0) -> List[str]:
    """Return ``lines`` passwords.  ``stream`` selects an independent draw from the
    same vocabulary (0 = training set, 1 = test set, ...)."""
    rng_v = np.random.default_rng(seed)
    words = _vocabulary(rng_v, vocab)
    p = 1.0 / np.arange(1, vocab + 1) ** s
    p /= p.sum()
    rng = np.random.default_rng([seed, 1000 + stream])
    idx = rng.choice(vocab, size=lines, p=p)
    out = [words[i] for i in idx]
    n_oov = int(round(lines * oov_frac))
    if n_oov:
        pos = rng.choice(lines, size=n_oov, replace=False)
        chars = np.array(list(ALPHABET + "ABCXYZ~"))
        for q in pos:
            ln = int(rng.integers(4, 15))
            out[q] = "".join(rng.choice(chars, size=ln))
    return out


def as_file(lines: List[str], name: str = "<synthetic>") -> io.StringIO:
    """A seekable text 'file' the trainers and parse_file accept (they need
    seek/read/iteration/close, as the reference's TextIO arguments do)."""
    f = io.StringIO("\n".join(lines) + "\n")
    f.name = name  # type: ignore[attr-defined]
    return f


def subword_corpus(lines: int, vocab: int = 3000, seed: int = DEFAULT_SEED, stream: int = 0,
                   sep: str = "\t") -> List[str]:
    """Lines of ``sep``-separated segments (for --splitter tab style models): each
    password is 1-3 vocabulary chunks so that several segmentations compete."""
    rng_v = np.random.default_rng(seed + 7)
    chunks = _vocabulary(rng_v, vocab)
    chunks = [c[: 2 + (i % 4)] for i, c in enumerate(chunks)]
    p = 1.0 / np.arange(1, vocab + 1) ** 1.05
    p /= p.sum()
    rng = np.random.default_rng([seed, 2000 + stream])
    k = rng.integers(1, 4, size=lines)
    out = []
    for n in k:
        idx = rng.choice(vocab, size=int(n), p=p)
        out.append(sep.join(chunks[i] for i in idx))
    return out


def v2_corpus(lines: int, vocab: int = 3000, seed: int = DEFAULT_SEED, stream: int = 0) -> List[str]:
    """nwords_v2 training format ``pwd\\tseg\\ttag\\tseg\\ttag...`` (reference
    nwords_v2/nwords_trainer_v2.py:21-23 reads items[0] and items[1::2])."""
    segs = subword_corpus(lines, vocab=vocab, seed=seed, stream=stream, sep="\x01")
    out = []
    for ln in segs:
        parts = ln.split("\x01")
        pwd = "".join(parts)
        if len(pwd) < 3:
            parts = parts + ["pad1"]
            pwd = "".join(parts)
        fields = [pwd]
        for sgm in parts:
            fields.append(sgm)
            fields.append("L" if sgm.isalpha() else "M")
        out.append("\t".join(fields))
    return out


def zipf_bpe_dir(path: str, rng: np.random.Generator, mixed: bool = False, structures: int = 400) -> None:
    """A synthetic BPE-PCFG model directory (format of the reference's bpeX/modelreader.py:15-68) with
    Zipfian terminals: rows of 10..1e5 terminals, ``structures`` structures (config 4's shape).  With
    ``mixed`` the directory also holds mixed-class terminal rows (mixed_2/3/4 = tags DM/TM/FM) and
    structures that use them, which is what sends ``calc_ml2p`` through the wildcard labels and the
    not_parsed fallback (bpe_simulator.py:72-101,121-128)."""
    import os
    for sub in ("grammar", "lower", "upper", "mixed_2", "mixed_3", "mixed_4", "digits", "special"):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    alpha = {"lower": "abcdefghijklmnopqrstuvwxyz", "upper": "ABCDEFGHIJKLMNOPQRSTUVWXYZ", "digits": "0123456789",
             "special": "!@#$%^&*._"}
    tag = {"lower": "L", "upper": "U", "digits": "D", "special": "S", "mixed_2": "DM", "mixed_3": "TM",
           "mixed_4": "FM"}
    sizes = {"lower": {1: 26, 2: 600, 3: 8000, 4: 60000, 5: 100000, 6: 100000, 7: 50000, 8: 30000},
             "upper": {1: 26, 2: 300, 3: 2000}, "digits": {1: 10, 2: 100, 3: 1000, 4: 10000, 5: 30000, 6: 50000},
             "special": {1: 10, 2: 90}}
    if mixed:
        both = alpha["lower"] + alpha["digits"]
        alpha.update({"mixed_2": both, "mixed_3": both + alpha["upper"], "mixed_4": both + alpha["special"]})
        sizes.update({"mixed_2": {2: 300, 3: 3000, 4: 20000}, "mixed_3": {3: 2000, 5: 20000},
                      "mixed_4": {4: 5000, 6: 40000}})
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
    while len(structs) < structures:
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
