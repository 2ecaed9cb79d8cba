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
