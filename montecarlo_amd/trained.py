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
    def __init__(self, handle, v2: bool = False):
        self._h = handle
        self.v2 = v2                 # mc_train_v2: contexts are strings of characters (code points)
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
                if self.v2:
                    ctx = "".join(map(chr, units[ctx_off[r]:ctx_off[r + 1]]))
                else:
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

    @classmethod
    def from_plain(cls, counts: Dict, words: Dict[str, int], start_chr: str, end_chr: str) -> Optional["NativeCounts"]:
        """The reference's plain ``{context: {symbol: count}}`` + ``words`` (what a model pickle holds,
        backwords_secondary_main.py:28-32) as native counts (mc_counts_import), or None when the dicts
        are not what the secondary trainer builds (the row of the empty context must be ``words``
        without the start symbol, counts must be non-negative ints): the caller then keeps the dicts."""
        start_cp, end_cp = one_code_point(start_chr), one_code_point(end_chr)
        if start_cp is None or end_cp is None or not isinstance(counts, dict) or not isinstance(words, dict):
            return None
        zero = {s: c for s, c in words.items() if s != start_chr}
        have = counts.get((), {} if not words else None)
        if have is None or list(have.items()) != list(zero.items()):
            return None
        sym_id: Dict[str, int] = {}
        try:
            for s in words:
                sym_id.setdefault(s, len(sym_id))
            rows = [(ctx, row) for ctx, row in counts.items() if ctx != ()]
            for ctx, row in rows:
                for s in ctx:
                    sym_id.setdefault(s, len(sym_id))
                for s in row:
                    sym_id.setdefault(s, len(sym_id))
            if not all(isinstance(s, str) for s in sym_id):
                return None
            text = "".join(sym_id).encode("utf-32-le", "surrogatepass")
            text_off = np.zeros(len(sym_id) + 1, dtype=np.uint32)
            np.cumsum(np.fromiter((len(s) for s in sym_id), dtype=np.uint32, count=len(sym_id)), out=text_off[1:])
            a = {
                "text_off": text_off, "text": np.frombuffer(text + b"\0\0\0\0", dtype="<u4"),
                "word_sym": np.fromiter((sym_id[s] for s in words), dtype=np.uint32, count=len(words)),
                "word_cnt": np.fromiter(words.values(), dtype=np.uint64, count=len(words)),
                "ctx_off": np.zeros(len(rows) + 1, dtype=np.uint32),
                "row_off": np.zeros(len(rows) + 1, dtype=np.uint64),
            }
            np.cumsum(np.fromiter((len(c) for c, _ in rows), dtype=np.uint32, count=len(rows)), out=a["ctx_off"][1:])
            np.cumsum(np.fromiter((len(r) for _, r in rows), dtype=np.uint64, count=len(rows)), out=a["row_off"][1:])
            a["ctx_units"] = np.fromiter((sym_id[s] for c, _ in rows for s in c), dtype=np.uint32,
                                         count=int(a["ctx_off"][-1]))
            a["ent_sym"] = np.fromiter((sym_id[s] for _, r in rows for s in r), dtype=np.uint32,
                                       count=int(a["row_off"][-1]))
            a["ent_cnt"] = np.fromiter((v for _, r in rows for v in r.values()), dtype=np.uint64,
                                       count=int(a["row_off"][-1]))
        except (TypeError, ValueError, OverflowError, AttributeError):
            return None
        out = cls()
        vp = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = nv.lib.mc_counts_import(out._h, vp(a["text_off"]), vp(a["text"]), len(sym_id), vp(a["word_sym"]),
                                     vp(a["word_cnt"]), len(words), vp(a["ctx_off"]), vp(a["ctx_units"]),
                                     vp(a["row_off"]), vp(a["ent_sym"]), vp(a["ent_cnt"]), len(rows), start_cp, end_cp)
        return out if rc == 0 else None

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
