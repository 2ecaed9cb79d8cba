"""Flatten a trained model (the reference's dict-of-dict form) into the arrays
mc_table_compile takes, and hold the compiled table.

Input is what the reference keeps in ``self.nwords`` after ``expand_2d``
(lib4mc/ProbLib.py:10-28): ``{context: (value dict, key list, fp64 cumsum)}`` in
dict order.  Rows and keys keep that order -- it defines the CDF the sampler
inverts.  ``-log2 p`` is computed here with ``math.log2`` exactly as the
reference's ``minus_log2`` does (lib4mc/MonteCarloParent.py:13), so the device
only adds precomputed fp64 values in draw order.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import _native as nv

STATE_DT = np.dtype([("total", "<f8"), ("ent_off", "<u4"), ("len", "<u4"), ("guide_off", "<u4"), ("gbits", "<u4"),
                     ("parent", "<u4"), ("ctx_len", "<u4")])
PAYLOAD_DT = np.dtype([("ml2p", "<f8"), ("next", "<u4"), ("sym", "<u4")])
SLOT_DT = np.dtype([("key", "<u8"), ("val", "<u4"), ("pad", "<u4")])
DRAW_DT = np.dtype([("w", "<u4", (4,))])


def _ptr(arr: np.ndarray, typ):
    return arr.ctypes.data_as(typ)


class FlatTable:
    """A compiled walk automaton in host memory (mc_table*) plus the id maps."""

    def __init__(self, expanded: Dict, kind: int, *, order: int = 0, start_units: Sequence[str] = (),
                 end_chr: str = "\x03", min_len: int = 4, max_len: int = 256, flags: int = 0,
                 minus_log_based: bool = False, unit_of_symbol=None, guide_factor: Optional[int] = None):
        if guide_factor is None:
            guide_factor = int(os.environ.get("MC_GUIDE_FACTOR", "4"))
        self.kind = kind
        sym_id: Dict[str, int] = {}
        symbols: List[str] = []

        def sid(s: str) -> int:
            i = sym_id.get(s)
            if i is None:
                i = sym_id[s] = len(symbols)
                symbols.append(s)
            return i

        split_units = kind == nv.KIND_V2
        unit_id = {} if split_units else sym_id        # v2: contexts are characters, transitions are words
        units: List[str] = [] if split_units else symbols

        def uid(u: str) -> int:
            if not split_units:
                return sid(u)
            i = unit_id.get(u)
            if i is None:
                i = unit_id[u] = len(units)
                units.append(u)
            return i

        n_ctx = len(expanded)
        ctx_off = np.zeros(n_ctx + 1, dtype=np.uint32)
        row_off = np.zeros(n_ctx + 1, dtype=np.uint64)
        ctx_units: List[int] = []
        ent_sym: List[int] = []
        ent_ml2p: List[float] = []
        cums: List[np.ndarray] = []
        self.ctx_row: Dict = {}
        for r, (ctx, (vals, keys, cum)) in enumerate(expanded.items()):
            self.ctx_row[ctx] = r
            ctx_units.extend(uid(u) for u in ctx)
            ctx_off[r + 1] = len(ctx_units)
            ent_sym.extend(sid(k) for k in keys)
            if minus_log_based:
                ent_ml2p.extend(vals[k] for k in keys)
            else:
                ent_ml2p.extend(-math.log2(vals[k]) for k in keys)
            cums.append(np.asarray(cum, dtype=np.float64))
            row_off[r + 1] = len(ent_sym)
        end_sym = sid(end_chr)
        start_ids = np.array([uid(u) for u in start_units], dtype=np.uint32)
        self.symbols = symbols
        self.sym_id = sym_id
        self.units = units
        self.unit_id = unit_id
        self.end_chr = end_chr
        self.end_sym = end_sym

        a_ctx_units = np.array(ctx_units, dtype=np.uint32)
        a_ent_sym = np.array(ent_sym, dtype=np.uint32)
        a_cum = np.concatenate(cums) if cums else np.zeros(0)
        a_ml2p = np.array(ent_ml2p, dtype=np.float64)
        a_chars = np.array([len(s) for s in symbols], dtype=np.uint32)
        keep = [ctx_off, row_off, a_ctx_units, a_ent_sym, a_cum, a_ml2p, a_chars, start_ids]

        d = nv.ModelDesc()
        d.kind, d.order, d.n_ctx, d.n_sym, d.n_ent = kind, order, n_ctx, len(symbols), len(ent_sym)
        d.ctx_off, d.ctx_units = _ptr(ctx_off, nv.u32p), _ptr(a_ctx_units, nv.u32p)
        d.row_off, d.ent_sym = _ptr(row_off, nv.u64p), _ptr(a_ent_sym, nv.u32p)
        d.ent_cum, d.ent_ml2p = _ptr(a_cum, nv.f64p), _ptr(a_ml2p, nv.f64p)
        if split_units:
            su_off = np.zeros(len(symbols) + 1, dtype=np.uint32)
            su: List[int] = []
            for i, s in enumerate(symbols):
                su.extend(uid(ch) for ch in s)
                su_off[i + 1] = len(su)
            a_su = np.array(su, dtype=np.uint32)
            keep += [su_off, a_su]
            d.sym_unit_off, d.sym_units = _ptr(su_off, nv.u32p), _ptr(a_su, nv.u32p)
        d.sym_chars = _ptr(a_chars, nv.u32p)
        text_off = np.zeros(len(symbols) + 1, dtype=np.uint32)
        np.cumsum(a_chars, out=text_off[1:])
        a_text = np.frombuffer("".join(symbols).encode("utf-32-le", "surrogatepass"), dtype="<u4").astype(np.uint32)
        keep += [text_off, a_text]
        d.sym_text_off, d.sym_text = _ptr(text_off, nv.u32p), _ptr(a_text, nv.u32p)
        d.start_units, d.n_start = _ptr(start_ids, nv.u32p), len(start_ids)
        d.end_sym, d.min_len, d.max_len, d.flags, d.guide_factor = end_sym, min_len, max_len, flags, guide_factor
        handle = C.c_void_p()
        nv.check(nv.lib.mc_table_compile(C.byref(d), C.byref(handle)))
        del keep
        self._h = handle
        info = (C.c_uint64 * 16)()
        nv.check(nv.lib.mc_table_info(self._h, info))
        self.info = list(info)
        self.multichar = bool(self.info[nv.INFO_MULTICHAR] & 1)
        self.max_word_len = int(self.info[nv.INFO_MULTICHAR] >> 8)
        self._lut: Optional[np.ndarray] = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and nv is not None and getattr(nv, "lib", None) is not None:   # (module may be gone at exit)
            nv.lib.mc_table_free(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    # ---- introspection (tests, DESIGN.md byte accounting) -------------------------------
    def array(self, what: int) -> np.ndarray:
        dt = {nv.ARR_STATE: STATE_DT, nv.ARR_DRAW: DRAW_DT, nv.ARR_CUM: np.dtype("<f8"),
              nv.ARR_PAYLOAD: PAYLOAD_DT, nv.ARR_TRANS: SLOT_DT, nv.ARR_CTXHASH: SLOT_DT,
              nv.ARR_SYMCHARS: np.dtype("<u4"), nv.ARR_SPAY: DRAW_DT, nv.ARR_RCUM: np.dtype("<f8"),
              nv.ARR_WORDHASH: SLOT_DT, nv.ARR_SYMTEXTOFF: np.dtype("<u4"), nv.ARR_SYMTEXT: np.dtype("<u4")}[what]
        size = nv.check(nv.lib.mc_table_array(self._h, what, None, 0))
        out = np.empty(size // dt.itemsize, dtype=dt)
        nv.check(nv.lib.mc_table_array(self._h, what, out.ctypes.data_as(C.c_void_p), size))
        return out

    # ---- host-side encode/decode of strings -----------------------------------------------
    def encode_chars(self, pwds: Iterable[str]):
        """Passwords -> (symbol ids, offsets) for single-character alphabets."""
        if self._lut is None:
            lut = np.full(0x110000, nv.MC_NONE, dtype=np.uint32)
            for s, i in self.sym_id.items():
                if len(s) == 1:
                    lut[ord(s)] = i
            self._lut = lut
        pwds = list(pwds)
        lens = np.fromiter((len(p) for p in pwds), dtype=np.uint64, count=len(pwds))
        off = np.zeros(len(pwds) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        cps = np.frombuffer("".join(pwds).encode("utf-32-le", "surrogatepass"), dtype="<u4")
        return self._lut[cps], off

    @staticmethod
    def encode_text(pwds: Iterable[str]):
        """Passwords -> (code points, offsets) for mc_score_words."""
        pwds = list(pwds)
        lens = np.fromiter((len(p) for p in pwds), dtype=np.uint64, count=len(pwds))
        off = np.zeros(len(pwds) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        cps = np.frombuffer("".join(pwds).encode("utf-32-le", "surrogatepass"), dtype="<u4").astype(np.uint32)
        return cps, off, int(lens.max()) if len(pwds) else 0

    def decode(self, syms: np.ndarray, lens: np.ndarray, drop: str = "") -> List[str]:
        """Symbol-id rows -> strings (``drop``: the start symbol, which the reference
        filters out of the returned string, nwords_simulator.py:116)."""
        table = self.symbols
        out = []
        for row, n in zip(syms, lens):
            parts = [table[i] for i in row[:n]]
            if drop:
                parts = [c for c in parts if c != drop]
            out.append("".join(parts))
        return out
