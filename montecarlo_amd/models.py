"""Glue shared by the simulator classes: attach a compiled table to a
``MonteCarlo`` instance and implement ``sample1`` / ``calc_ml2p`` /
``score_batch`` on top of the device kernels."""
from __future__ import annotations

from typing import List, Tuple

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
        flat = self._flat
        if flat.multichar:
            raise NotImplementedError("scoring of multi-character (sub-word) alphabets is not built yet")
        syms, off = flat.encode_chars(pwds)
        ml2p = self._dev.score_ids(syms, off).cpu().numpy().tolist()
        end = self.end_chr
        parts = [[c for c in p if c != end] if m < UNPARSEABLE else ([p] if p != end else [])
                 for p, m in zip(pwds, ml2p)]
        return ml2p, parts
