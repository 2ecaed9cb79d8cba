"""A model trained by the native trainers (csrc/mc_train.cpp): the flattened rows stay in C++
and go straight into the table compiler; the reference's dict-of-dict view (``nwords`` after
``expand_2d``, ``words``) is rebuilt from them only when somebody reads it."""
from __future__ import annotations

import ctypes as C
import os
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
