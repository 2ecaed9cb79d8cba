"""A model trained by the native trainers (csrc/mc_train.cpp): the flattened rows stay in C++
and go straight into the table compiler; the reference's dict-of-dict view (``nwords`` after
``expand_2d``, ``words``) is rebuilt from them only when somebody reads it."""
from __future__ import annotations

import ctypes as C
import os
import collections.abc
from typing import Dict, List, Optional

import numpy as np

from . import _native as nv
from .tables import FlatTable

_DT = {nv.TRAINED_CTX_OFF: "<u4", nv.TRAINED_CTX_UNITS: "<u4", nv.TRAINED_ROW_OFF: "<u8", nv.TRAINED_ENT_SYM: "<u4",
       nv.TRAINED_ENT_PROB: "<f8", nv.TRAINED_ENT_CUM: "<f8", nv.TRAINED_ENT_ML2P: "<f8",
       nv.TRAINED_SYM_TEXT_OFF: "<u4", nv.TRAINED_SYM_TEXT: "<u4", nv.TRAINED_WORD_SYM: "<u4",
       nv.TRAINED_WORD_COUNT: "<u8"}


def one_code_point(s: str) -> Optional[int]:
    return ord(s) if isinstance(s, str) and len(s) == 1 else None


def splitter_code_point(splitter: str) -> Optional[int]:
    """0 for '' (every character a symbol), the code point of a one-character LITERAL separator,
    None for anything the native trainers do not take (``re.split`` patterns)."""
    if splitter == "":
        return 0
    if len(splitter) == 1 and (splitter.isalnum() or splitter in " \t,;:/_-#@=&%~!\"'<>`"):
        return ord(splitter)
    return None


def read_text(source) -> str:
    """The decoded training text; a file is closed afterwards as the reference's trainers do
    (nwords_trainer.py:39)."""
    if hasattr(source, "read"):
        try:
            if source.seekable():
                source.seek(0)
            return source.read()
        finally:
            source.close()
    return "".join(line if line.endswith("\n") else line + "\n" for line in source)


class Trained:
    def __init__(self, handle):
        self._h = handle
        off = self.array(nv.TRAINED_SYM_TEXT_OFF)
        text = self.array(nv.TRAINED_SYM_TEXT).astype("<u4").tobytes().decode("utf-32-le", "surrogatepass")
        self.symbols: List[str] = [text[a:b] for a, b in zip(off[:-1].tolist(), off[1:].tolist())]
        self.words: Dict[str, int] = {self.symbols[s]: int(c) for s, c in
                                      zip(self.array(nv.TRAINED_WORD_SYM).tolist(),
                                          self.array(nv.TRAINED_WORD_COUNT).tolist())}
        self._nwords = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and nv is not None and getattr(nv, "lib", None) is not None:
            nv.lib.mc_trained_free(h)
            self._h = None

    def array(self, what: int) -> np.ndarray:
        size = nv.check(nv.lib.mc_trained_array(self._h, what, None, 0))
        dt = np.dtype(_DT[what])
        out = np.empty(size // dt.itemsize, dtype=dt)
        if size:
            nv.check(nv.lib.mc_trained_array(self._h, what, out.ctypes.data_as(C.c_void_p), size))
        return out

    @property
    def nwords(self):
        """``expand_2d(trainer output)``: {context tuple: ({symbol: p}, [symbols], fp64 cumsum)}."""
        if self._nwords is None:
            sym = self.symbols
            ctx_off, units = self.array(nv.TRAINED_CTX_OFF).tolist(), self.array(nv.TRAINED_CTX_UNITS).tolist()
            row_off = self.array(nv.TRAINED_ROW_OFF).tolist()
            ent_sym, prob = self.array(nv.TRAINED_ENT_SYM).tolist(), self.array(nv.TRAINED_ENT_PROB).tolist()
            cum = self.array(nv.TRAINED_ENT_CUM)
            table = {}
            for r in range(len(ctx_off) - 1):
                ctx = tuple(sym[u] for u in units[ctx_off[r]:ctx_off[r + 1]])
                a, b = row_off[r], row_off[r + 1]
                keys = [sym[s] for s in ent_sym[a:b]]
                table[ctx] = (dict(zip(keys, prob[a:b])), keys, cum[a:b].copy())
            self._nwords = table
        return self._nwords

    def flat(self, kind: int, *, order: int, end_chr: str, min_len: int = 4, max_len: int = 256, flags: int = 0,
             guide_factor: Optional[int] = None) -> "FlatTable":
        if guide_factor is None:
            guide_factor = int(os.environ.get("MC_GUIDE_FACTOR", "2"))
        handle = C.c_void_p()
        nv.check(nv.lib.mc_trained_compile(self._h, kind, order, min_len, max_len, flags, guide_factor,
                                           C.byref(handle)))
        return FlatTable.adopt(handle, kind, self.symbols, end_chr)


class NativeCounts(collections.abc.Mapping):
    """The secondary trainer's raw counts ``{context: {symbol: count}}`` held in C++ (mc_counts): merged
    round by round and turned into probabilities there.  Read as a mapping (or pickled) it shows the
    reference's plain dict-of-dict, rebuilt from the native arrays after every merge; pickling
    writes exactly that plain dict, so the files stay loadable without this package."""

    def __init__(self):
        h = C.c_void_p()
        nv.check(nv.lib.mc_counts_new(C.byref(h)))
        self._h = h
        self._view = None
        self.signature = None        # (splitter_cp, start_cp, end_cp) of the merges so far

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and nv is not None and getattr(nv, "lib", None) is not None:
            nv.lib.mc_counts_free(h)
            self._h = None

    def merge(self, text: str, splitter_cp: int, start4word: int, skip4word: int, max_gram: int, threshold: int,
              start_cp: int, end_cp: int):
        data = text.encode("utf-8", "surrogatepass")
        nv.check(nv.lib.mc_counts_merge(self._h, data, len(data), splitter_cp, start4word, skip4word,
                                        min(max_gram, 0xFFFFFFFF), threshold, start_cp, end_cp))
        self.signature = (splitter_cp, start_cp, end_cp)
        self._view = None

    def array(self, what: int) -> np.ndarray:
        size = nv.check(nv.lib.mc_counts_array(self._h, what, None, 0))
        dt = np.dtype("<u8" if what in (nv.TRAINED_ROW_OFF, nv.TRAINED_WORD_COUNT, nv.COUNTS_ENT_COUNT) else "<u4")
        out = np.empty(size // dt.itemsize, dtype=dt)
        if size:
            nv.check(nv.lib.mc_counts_array(self._h, what, out.ctypes.data_as(C.c_void_p), size))
        return out

    def symbols(self) -> List[str]:
        off = self.array(nv.TRAINED_SYM_TEXT_OFF)
        text = self.array(nv.TRAINED_SYM_TEXT).astype("<u4").tobytes().decode("utf-32-le", "surrogatepass")
        return [text[a:b] for a, b in zip(off[:-1].tolist(), off[1:].tolist())]

    def words(self) -> Dict[str, int]:
        sym = self.symbols()
        return {sym[s]: int(c) for s, c in zip(self.array(nv.TRAINED_WORD_SYM).tolist(),
                                               self.array(nv.TRAINED_WORD_COUNT).tolist())}

    def plain(self) -> Dict:
        if self._view is None:
            sym = self.symbols()
            ctx_off, units = self.array(nv.TRAINED_CTX_OFF).tolist(), self.array(nv.TRAINED_CTX_UNITS).tolist()
            row_off = self.array(nv.TRAINED_ROW_OFF).tolist()
            ent_sym, cnt = self.array(nv.TRAINED_ENT_SYM).tolist(), self.array(nv.COUNTS_ENT_COUNT).tolist()
            view = {}
            for r in range(len(ctx_off) - 1):
                a, b = row_off[r], row_off[r + 1]
                view[tuple(sym[u] for u in units[ctx_off[r]:ctx_off[r + 1]])] = \
                    {sym[s]: c for s, c in zip(ent_sym[a:b], cnt[a:b])}
            self._view = view
        return self._view

    def train(self, threshold: int) -> Trained:
        h = C.c_void_p()
        nv.check(nv.lib.mc_counts_train(self._h, threshold, C.byref(h)))
        return Trained(h)

    def __getitem__(self, key):
        return self.plain()[key]

    def __iter__(self):
        return iter(self.plain())

    def __len__(self):
        return len(self.plain())

    def __reduce__(self):
        return dict, (self.plain(),)
