"""Test helper: a plain-Python reading of the COMPILED table (the arrays
mc_table_compile produced), stepping exactly as the HIP kernels do.  It lets the
CPU test-suite check the table compiler and the walk design against the
reference's golden vectors in a container without a GPU.  It is not a product
path and nothing under montecarlo_amd/ imports it."""
from __future__ import annotations

import numpy as np

from montecarlo_amd import _native as nv

NONE = nv.MC_NONE
MASK64 = (1 << 64) - 1


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


class CompiledWalker:
    def __init__(self, flat):
        self.flat = flat
        self.state = flat.array(nv.ARR_STATE)
        self.guide = flat.array(nv.ARR_GUIDE)
        self.cum = flat.array(nv.ARR_CUM)
        self.payload = flat.array(nv.ARR_PAYLOAD)
        self.trans = flat.array(nv.ARR_TRANS)
        self.sym_chars = flat.array(nv.ARR_SYMCHARS)
        self.start = flat.info[nv.INFO_START_STATE]

    # mirrors k_sample: returns (ml2p, symbol ids, draws)
    def sample(self, uniform, min_len=4, max_len=256, keep_len=False, max_draws=1 << 16):
        st, acc, plen, syms, draw = self.start, 0.0, 0, [], 0
        while True:
            r = uniform()
            rec = self.state[st]
            K = 1 << int(rec["gbits"])
            k = min(int(r * K), K - 1)
            lo = int(self.guide[int(rec["guide_off"]) + k])
            hi = int(self.guide[int(rec["guide_off"]) + k + 1])
            x = float(rec["total"]) * r
            while lo < hi:
                mid = (lo + hi) >> 1
                if self.cum[mid] > x:
                    hi = mid
                else:
                    lo = mid + 1
            p = self.payload[lo]
            acc += float(p["ml2p"])
            draw += 1
            restart = False
            if int(p["sym"]) & 0x80000000:
                if plen >= min_len:
                    return acc, syms, draw
                restart = True
            else:
                s = int(p["sym"])
                syms.append(s)
                plen += int(self.sym_chars[s])
                st = int(p["next"])
                if plen >= max_len:
                    restart = True
            if restart:
                st, acc, syms = self.start, 0.0, []
                if not keep_len:
                    plen = 0
            if draw >= max_draws or st == NONE:
                return float("nan"), syms, draw

    def find(self, row, s):
        key = (row << 32) | s
        mask = len(self.trans) - 1
        q = mix64((key + 0x9E3779B97F4A7C15) & MASK64) & mask
        while True:
            sl = self.trans[q]
            if int(sl["key"]) == key:
                return int(sl["val"])
            if int(sl["key"]) == MASK64:
                return NONE
            q = (q + 1) & mask

    # mirrors k_score
    def score(self, sym_ids, end_sym, n_sym):
        st, acc = self.start, 0.0
        for s in list(sym_ids) + [end_sym]:
            if s >= n_sym or st == NONE:
                return 1022.0
            row, e = st, NONE
            while row != NONE:
                e = self.find(row, s)
                if e != NONE:
                    break
                row = int(self.state[row]["parent"])
            if e == NONE:
                return 1022.0
            p = self.payload[e]
            acc += float(p["ml2p"])
            st = int(p["next"])
        return acc if acc < 1022.0 else 1022.0
