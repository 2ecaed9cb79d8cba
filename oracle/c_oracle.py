"""ctypes front end of oracle/mc_oracle.c -- TEST INFRASTRUCTURE ONLY.

Flattens an expanded table (``{ctx: (vals, keys, cum)}``, the reference's
``expand_2d`` output) with its OWN symbol numbering and drives the C restatement.
Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never
imported from montecarlo_amd/.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import Dict, List, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmc_oracle.so")
KIND_NGRAM, KIND_BACKOFF, KIND_V2 = 1, 2, 3
NONE = 0xFFFFFFFF

u32p, u64p, f64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_double)


class Desc(C.Structure):           # = struct mc_model_desc, include/mi355mc.h
    _fields_ = [
        ("kind", C.c_uint32), ("order", C.c_uint32), ("n_ctx", C.c_uint32), ("n_sym", C.c_uint32),
        ("n_ent", C.c_uint64),
        ("ctx_off", u32p), ("ctx_units", u32p), ("row_off", u64p), ("ent_sym", u32p),
        ("ent_cum", f64p), ("ent_ml2p", f64p),
        ("sym_unit_off", u32p), ("sym_units", u32p), ("sym_chars", u32p),
        ("start_units", u32p), ("n_start", C.c_uint32),
        ("end_sym", C.c_uint32), ("min_len", C.c_uint32), ("max_len", C.c_uint32), ("flags", C.c_uint32),
        ("guide_factor", C.c_uint32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "mc_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B", "libmc_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        vp = C.c_void_p
        L.orc_model_new.restype, L.orc_model_new.argtypes = vp, [C.POINTER(Desc)]
        L.orc_model_free.restype, L.orc_model_free.argtypes = None, [vp]
        L.orc_uniform.restype, L.orc_uniform.argtypes = C.c_double, [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_sample.restype = C.c_int
        L.orc_sample.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint32]
        L.orc_score.restype, L.orc_score.argtypes = C.c_int, [vp, vp, vp, C.c_uint64, vp]
        L.orc_rank_build.restype, L.orc_rank_build.argtypes = None, [vp, C.c_uint64, C.c_uint64, vp, vp]
        L.orc_rank_query.restype = None
        L.orc_rank_query.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class COracleModel:
    def __init__(self, expanded: Dict, kind: int, *, order: int = 0, start_units: Sequence[str] = (),
                 end_chr: str = "\x03", min_len: int = 4, max_len: int = 256, flags: int = 0):
        sym_id: Dict[str, int] = {}
        self.symbols: List[str] = []

        def sid(s):
            if s not in sym_id:
                sym_id[s] = len(self.symbols)
                self.symbols.append(s)
            return sym_id[s]

        v2 = kind == KIND_V2
        unit_id: Dict[str, int] = {} if v2 else sym_id

        def uid(u):
            if not v2:
                return sid(u)
            return unit_id.setdefault(u, len(unit_id))

        ctx_off, row_off, ctx_units, ent_sym, ml2p, cums = [0], [0], [], [], [], []
        for ctx, (vals, keys, cum) in expanded.items():
            ctx_units += [uid(u) for u in ctx]
            ctx_off.append(len(ctx_units))
            ent_sym += [sid(k) for k in keys]
            ml2p += [-math.log2(vals[k]) for k in keys]
            cums.append(np.asarray(cum, dtype=np.float64))
            row_off.append(len(ent_sym))
        self.end_sym = sid(end_chr)
        start = [uid(u) for u in start_units]
        self.sym_id = sym_id
        a = self._arrays = {
            "ctx_off": np.array(ctx_off, dtype=np.uint32), "ctx_units": np.array(ctx_units + [0], dtype=np.uint32),
            "row_off": np.array(row_off, dtype=np.uint64), "ent_sym": np.array(ent_sym, dtype=np.uint32),
            "ent_cum": np.concatenate(cums), "ent_ml2p": np.array(ml2p, dtype=np.float64),
            "sym_chars": np.array([len(s) for s in self.symbols], dtype=np.uint32),
            "start": np.array(start + [0], dtype=np.uint32),
        }
        d = Desc()
        d.kind, d.order, d.n_ctx, d.n_sym, d.n_ent = kind, order, len(ctx_off) - 1, len(self.symbols), len(ent_sym)
        for f, k in (("ctx_off", "ctx_off"), ("ctx_units", "ctx_units"), ("ent_sym", "ent_sym"),
                     ("sym_chars", "sym_chars"), ("start_units", "start")):
            setattr(d, f, a[k].ctypes.data_as(u32p))
        d.row_off = a["row_off"].ctypes.data_as(u64p)
        d.ent_cum, d.ent_ml2p = a["ent_cum"].ctypes.data_as(f64p), a["ent_ml2p"].ctypes.data_as(f64p)
        if v2:
            off, su = [0], []
            for s in self.symbols:
                su += [uid(ch) for ch in s]
                off.append(len(su))
            a["su_off"], a["su"] = np.array(off, dtype=np.uint32), np.array(su + [0], dtype=np.uint32)
            d.sym_unit_off, d.sym_units = a["su_off"].ctypes.data_as(u32p), a["su"].ctypes.data_as(u32p)
        d.n_start, d.end_sym, d.min_len, d.max_len, d.flags = len(start), self.end_sym, min_len, max_len, flags
        self._desc = d
        self._h = C.c_void_p(lib().orc_model_new(C.byref(d)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:  # (module globals go first at interpreter exit)
            _lib.orc_model_free(self._h)
            self._h = None

    def sample(self, count: int, seed: int, first_id: int = 0, max_draws: int = 1 << 16, stride: int = 0,
               uniforms: np.ndarray = None):
        ml2p = np.empty(count, dtype=np.float64)
        ln = np.empty(count, dtype=np.uint32)
        dr = np.empty(count, dtype=np.uint32)
        sy = np.zeros((count, max(stride, 1)), dtype=np.uint32)
        u = None
        if uniforms is not None:
            u = np.ascontiguousarray(uniforms, dtype=np.float64)
            max_draws = u.shape[1]
        lib().orc_sample(self._h, first_id, count, seed, max_draws, _p(u) if u is not None else None, _p(ml2p),
                         _p(ln), _p(dr), _p(sy) if stride else None, stride)
        return {"ml2p": ml2p, "len": ln, "draws": dr, "syms": sy if stride else None}

    def strings(self, res, drop: str = "") -> List[str]:
        out = []
        for row, n in zip(res["syms"], res["len"]):
            out.append("".join(c for c in (self.symbols[i] for i in row[:n]) if c != drop))
        return out

    def score(self, pwds: Sequence[str]) -> np.ndarray:
        """Single-character alphabets only (one parse per password)."""
        ids, off = [], [0]
        for p in pwds:
            ids += [self.sym_id.get(ch, NONE) for ch in p]
            off.append(len(ids))
        a_ids, a_off = np.array(ids + [0], dtype=np.uint32), np.array(off, dtype=np.uint64)
        out = np.empty(len(pwds), dtype=np.float64)
        lib().orc_score(self._h, _p(a_ids), _p(a_off), len(pwds), _p(out))
        return out


def rank_build(keys: np.ndarray, n_global: int = None):
    keys = np.ascontiguousarray(keys, dtype=np.float64)
    srt, pos = np.empty_like(keys), np.empty_like(keys)
    lib().orc_rank_build(_p(keys), len(keys), n_global or len(keys), _p(srt), _p(pos))
    return srt, pos


def rank_query(srt: np.ndarray, pos: np.ndarray, x: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out_pos, out_cnt = np.empty(len(x), dtype=np.float64), np.empty(len(x), dtype=np.uint64)
    lib().orc_rank_query(_p(srt), _p(pos), len(srt), _p(x), len(x), _p(out_pos), _p(out_cnt))
    return out_pos, out_cnt
