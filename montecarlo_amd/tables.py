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
            guide_factor = int(os.environ.get("MC_GUIDE_FACTOR", "2"))
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
        self.len_in_rec = bool(self.info[nv.INFO_MULTICHAR] & 2)   # (the draw records carry the symbols' lengths)
        self.max_word_len = int(self.info[nv.INFO_MULTICHAR] >> 8)
        self._lut: Optional[np.ndarray] = None

    @classmethod
    def adopt(cls, handle, kind: int, symbols: List[str], end_chr: str) -> "FlatTable":
        """Wrap a table compiled elsewhere (mc_trained_compile) whose symbol ids are ``symbols``."""
        self = cls.__new__(cls)
        self.kind = kind
        self._h = handle
        self.symbols = list(symbols)
        self.sym_id = {s: i for i, s in enumerate(symbols)}
        self.units, self.unit_id = self.symbols, self.sym_id
        self.end_chr, self.end_sym = end_chr, self.sym_id[end_chr]
        self.ctx_row = {}
        info = (C.c_uint64 * 16)()
        nv.check(nv.lib.mc_table_info(self._h, info))
        self.info = list(info)
        self.multichar = bool(self.info[nv.INFO_MULTICHAR] & 1)
        self.len_in_rec = bool(self.info[nv.INFO_MULTICHAR] & 2)   # (the draw records carry the symbols' lengths)
        self.max_word_len = int(self.info[nv.INFO_MULTICHAR] >> 8)
        self._lut = None
        return self

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
              nv.ARR_WORDHASH: SLOT_DT, nv.ARR_SYMTEXTOFF: np.dtype("<u4"), nv.ARR_SYMTEXT: np.dtype("<u4"), nv.ARR_DENSE: PAYLOAD_DT,
              nv.ARR_WORDCK: np.dtype("<u8")}[what]
        size = nv.check(nv.lib.mc_table_array(self._h, what, None, 0))
        out = np.empty(size // dt.itemsize, dtype=dt)
        nv.check(nv.lib.mc_table_array(self._h, what, out.ctypes.data_as(C.c_void_p), size))
        return out

    # ---- host-side encode/decode of strings -----------------------------------------------
    def encode_chars(self, pwds: Iterable[str]):
        """Passwords -> (symbol ids, offsets) for single-character alphabets."""
        pwds = list(pwds)
        lens = np.fromiter((len(p) for p in pwds), dtype=np.uint64, count=len(pwds))
        off = np.zeros(len(pwds) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        cps = np.frombuffer("".join(pwds).encode("utf-32-le", "surrogatepass"), dtype="<u4")
        return self.char_lut()[cps], off

    def char_lut(self) -> np.ndarray:
        """code point -> symbol id (MC_NONE outside the alphabet), single-character symbols only."""
        if self._lut is None:
            lut = np.full(0x110000, nv.MC_NONE, dtype=np.uint32)
            for s, i in self.sym_id.items():
                if len(s) == 1:
                    lut[ord(s)] = i
            self._lut = lut
        return self._lut

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


class BpeFlat(FlatTable):
    """Two-level BPE-PCFG table (bpe_simulator.py:152-156): row 0 = the grammar (entry i = structure
    i, in ``expand_1d`` order), rows 1.. = the terminal rows of ``expand_2d`` in dict order.  A
    structure's segments name terminal rows; the label / not_parsed index of ``count_luds`` is
    flattened beside it for the scoring kernel."""

    def __init__(self, grammars, terminals, converted, not_parsed, class_of, guide_factor: Optional[int] = None):
        if guide_factor is None:
            guide_factor = int(os.environ.get("MC_GUIDE_FACTOR", "2"))
        g_vals, g_keys, g_cum = grammars
        self.kind = nv.KIND_BPE
        self.structures = list(g_keys)
        struct_id = {st: i for i, st in enumerate(self.structures)}
        row_of = {key: r + 1 for r, key in enumerate(terminals)}
        n_struct, n_ctx = len(self.structures), 1 + len(terminals)
        # symbols: one (textless) id per structure, one per distinct terminal string, a textless end marker
        symbols: List[str] = [""] * n_struct
        sym_id: Dict[str, int] = {}
        ent_sym: List[int] = list(range(n_struct))
        ent_ml2p: List[float] = [g_vals[k] for k in g_keys]
        cums = [np.asarray(g_cum, dtype=np.float64)]
        row_off = np.zeros(n_ctx + 1, dtype=np.uint64)
        row_off[1] = n_struct
        self.entry_text: List[str] = [""] * n_struct
        for r, (key, (vals, keys, cum)) in enumerate(terminals.items()):
            for k in keys:
                i = sym_id.get(k)
                if i is None:
                    i = sym_id[k] = len(symbols)
                    symbols.append(k)
                ent_sym.append(i)
                ent_ml2p.append(vals[k])
            self.entry_text.extend(keys)
            cums.append(np.asarray(cum, dtype=np.float64))
            row_off[r + 2] = len(ent_sym)
        end_sym = len(symbols)
        symbols.append("")
        self.symbols, self.sym_id, self.end_sym, self.end_chr = symbols, sym_id, end_sym, ""
        seg_off = np.zeros(n_struct + 1, dtype=np.uint32)
        seg_row: List[int] = []
        seg_len: List[int] = []
        for i, st in enumerate(self.structures):
            for tag, ln in st:
                seg_row.append(row_of.get((tag, ln), nv.MC_NONE))
                seg_len.append(ln)
            seg_off[i + 1] = len(seg_row)
        self.max_segments = max(1, int(np.diff(seg_off).max()) if n_struct else 1)
        labels = [lab for lab in converted if all(tag in class_of for tag, _ in lab)]   # luds only emits L/U/D/S
        lab_off = np.zeros(len(labels) + 1, dtype=np.uint32)
        lab_cand_off = np.zeros(len(labels) + 1, dtype=np.uint32)
        lab_units: List[int] = []
        lab_cand: List[int] = []
        for j, lab in enumerate(labels):
            lab_units.extend((class_of[tag] << 24) | ln for tag, ln in lab)
            lab_off[j + 1] = len(lab_units)
            lab_cand.extend(sorted(struct_id[st] for st in converted[lab]))
            lab_cand_off[j + 1] = len(lab_cand)
        np_len = np.array(list(not_parsed), dtype=np.uint32)
        np_off = np.zeros(len(not_parsed) + 1, dtype=np.uint32)
        np_cand: List[int] = []
        for k, total in enumerate(not_parsed):
            np_cand.extend(sorted(struct_id[st] for st in not_parsed[total]))
            np_off[k + 1] = len(np_cand)

        ctx_off = np.arange(n_ctx + 1, dtype=np.uint32)
        ctx_units = np.arange(n_ctx, dtype=np.uint32)
        a_ent_sym = np.array(ent_sym, dtype=np.uint32)
        a_cum = np.concatenate(cums)
        a_ml2p = np.array(ent_ml2p, dtype=np.float64)
        a_chars = np.array([len(s) for s in symbols], dtype=np.uint32)
        text_off = np.zeros(len(symbols) + 1, dtype=np.uint32)
        np.cumsum(a_chars, out=text_off[1:])
        a_text = np.frombuffer("".join(symbols).encode("utf-32-le", "surrogatepass"), dtype="<u4").astype(np.uint32)
        start_ids = np.zeros(1, dtype=np.uint32)
        a_seg_row, a_seg_len = np.array(seg_row, dtype=np.uint32), np.array(seg_len, dtype=np.uint32)
        a_lab_units, a_lab_cand = np.array(lab_units, dtype=np.uint32), np.array(lab_cand, dtype=np.uint32)
        a_np_cand = np.array(np_cand, dtype=np.uint32)
        keep = [ctx_off, ctx_units, row_off, a_ent_sym, a_cum, a_ml2p, a_chars, text_off, a_text, start_ids, seg_off,
                a_seg_row, a_seg_len, lab_off, lab_cand_off, a_lab_units, a_lab_cand, np_len, np_off, a_np_cand]

        d = nv.ModelDesc()
        d.kind, d.order, d.n_ctx, d.n_sym, d.n_ent = nv.KIND_BPE, 0, n_ctx, len(symbols), len(ent_sym)
        d.ctx_off, d.ctx_units = _ptr(ctx_off, nv.u32p), _ptr(ctx_units, nv.u32p)
        d.row_off, d.ent_sym = _ptr(row_off, nv.u64p), _ptr(a_ent_sym, nv.u32p)
        d.ent_cum, d.ent_ml2p = _ptr(a_cum, nv.f64p), _ptr(a_ml2p, nv.f64p)
        d.sym_chars = _ptr(a_chars, nv.u32p)
        d.start_units, d.n_start = _ptr(start_ids, nv.u32p), 1
        d.end_sym, d.min_len, d.max_len, d.flags, d.guide_factor = end_sym, 0, 1 << 30, 0, guide_factor
        d.sym_text_off, d.sym_text = _ptr(text_off, nv.u32p), _ptr(a_text, nv.u32p)
        d.seg_off, d.seg_row, d.seg_len = _ptr(seg_off, nv.u32p), _ptr(a_seg_row, nv.u32p), _ptr(a_seg_len, nv.u32p)
        d.n_labels, d.n_np = len(labels), len(np_len)
        d.lab_off, d.lab_units = _ptr(lab_off, nv.u32p), _ptr(a_lab_units, nv.u32p)
        d.lab_cand_off, d.lab_cand = _ptr(lab_cand_off, nv.u32p), _ptr(a_lab_cand, nv.u32p)
        d.np_len, d.np_off, d.np_cand = _ptr(np_len, nv.u32p), _ptr(np_off, nv.u32p), _ptr(a_np_cand, nv.u32p)
        handle = C.c_void_p()
        nv.check(nv.lib.mc_table_compile(C.byref(d), C.byref(handle)))
        del keep
        self._h = handle
        info = (C.c_uint64 * 16)()
        nv.check(nv.lib.mc_table_info(self._h, info))
        self.info = list(info)
        self.multichar = True
        self.max_word_len = int(self.info[nv.INFO_MULTICHAR] >> 8)
        self._lut = None
        self.ctx_row = dict(row_of)
        # what was handed to the compiler, for tests/walker.py
        self.seg = (seg_off.copy(), a_seg_row.copy(), a_seg_len.copy())
        self.label_index = {tuple((class_of[tag] << 24) | ln for tag, ln in lab):
                            lab_cand[lab_cand_off[j]:lab_cand_off[j + 1]] for j, lab in enumerate(labels)}
        self.np_index = {int(total): np_cand[np_off[k]:np_off[k + 1]] for k, total in enumerate(not_parsed)}

    def decode_entries(self, ents: np.ndarray, lens: np.ndarray) -> List[str]:
        """Rows of terminal ENTRY indices (k_sample_bpe's out_syms) -> sampled strings."""
        text = self.entry_text
        return ["".join(text[e] for e in row[:n]) for row, n in zip(ents, lens)]
