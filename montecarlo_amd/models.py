"""Glue shared by the simulator classes: attach a compiled table to a
``MonteCarlo`` instance and implement ``sample1`` / ``calc_ml2p`` /
``score_batch`` on top of the device kernels."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import _native as nv
from .tables import FlatTable

UNPARSEABLE = float(1022)


class TableBacked:
    """Mixin for models whose walk is the compiled automaton (n-gram, backoff, v2)."""

    def _attach(self, kind: int, *, order: int, start_units, end_chr: str, min_len: int = 4, flags: int = 0):
        self._flat = FlatTable(self.nwords, kind, order=order, start_units=start_units, end_chr=end_chr,
                               min_len=min_len, max_len=256, flags=flags)
        self._dev_model = None

    # sample1 (nwords_simulator.py:92-116): one id of the model's Philox stream
    def sample1(self) -> Tuple[float, str]:
        seed, first = self._claim_ids(1)
        res = self._dev.sample(1, seed, first_id=first, max_draws=self.max_draws, sym_stride=32)
        self._dev.check()
        pwd = self._strings_of(res, seed, first)[0]
        return float(res["ml2p"].cpu()[0]), pwd

    def calc_ml2p(self, pwd: str) -> Tuple[float, List[str]]:
        ml2p, parts = self.score_batch([pwd])
        return ml2p[0], parts[0]

    def score_batch(self, pwds: List[str]):
        pwds = list(pwds)
        if not self._flat.multichar:
            syms, off = self._flat.encode_chars(pwds)
            return self._score_chars(pwds, syms, off, True)
        return self.score_batch_dfs(pwds)

    def score_batch_dfs(self, pwds: List[str]):
        """calc_ml2p through the DFS kernel (sub-word alphabets; valid for any alphabet)."""
        if not pwds:
            return [], []
        cps, off, max_len = self._flat.encode_text(pwds)
        return self._score_words(pwds, cps, off, max_len, True)

    def _score_unique(self, lines, pwds: List[str], want_parts: bool):
        """score_batch on lines the native ingest has already encoded (montecarlo_amd.ingest)."""
        if not pwds:
            return [], []
        if not self._flat.multichar:
            return self._score_chars(pwds, self._flat.char_lut()[lines.cps], lines.off, want_parts)
        return self._score_words(pwds, lines.cps, lines.off, lines.longest, want_parts)

    def _score_unique_device(self, lines):
        """-log2 p of the ingest's unique lines as a device vector (parse_file without components: nothing
        but the order and the sorted keys ever comes back to the host)."""
        if not self._flat.multichar:
            return self._dev.score_ids(self._flat.char_lut()[lines.cps], lines.off)
        res, _ = self._dev.score_words(lines.cps, lines.off, lines.longest, getattr(self, "max_iter", 10 ** 35),
                                       want_split=False)
        self._dev.check()
        return res

    def _score_chars(self, pwds, syms, off, want_parts: bool):
        # single-character alphabets have exactly one parse: the lookup-chain kernel
        end = self.end_chr
        ml2p = self._dev.score_ids(syms, off).cpu().numpy().tolist()
        if not want_parts:
            return ml2p, None
        parts = [[c for c in p if c != end] if m < UNPARSEABLE else ([p] if p != end else [])
                 for p, m in zip(pwds, ml2p)]
        return ml2p, parts

    def _score_words(self, pwds, cps, off, max_len: int, want_parts: bool):
        iter_max = getattr(self, "max_iter", 10 ** 35)          # nwords_simulator.py:88 passes 1e35
        res, split = self._dev.score_words(cps, off, max_len, iter_max)
        self._dev.check()
        ml2p = res.cpu().numpy().tolist()
        if not want_parts:
            return ml2p, None
        masks = split.cpu().numpy().view(np.uint64)
        end = self.end_chr
        parts = []
        for p, m, row in zip(pwds, ml2p, masks):
            if not m < UNPARSEABLE:
                parts.append([p] if p != end else [])       # component_list starts as [pwd] (:86)
                continue
            text, toks, prev = p + end, [], 0
            for j in range(len(text)):
                if (int(row[j >> 6]) >> (j & 63)) & 1:
                    toks.append(text[prev:j + 1])
                    prev = j + 1
            parts.append([c for c in toks if c != end])
        return ml2p, parts
