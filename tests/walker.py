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
        self.draw = flat.array(nv.ARR_DRAW)["w"]
        self.spay = flat.array(nv.ARR_SPAY)["w"]
        self.rcum = flat.array(nv.ARR_RCUM)
        self.cum = flat.array(nv.ARR_CUM)
        self.payload = flat.array(nv.ARR_PAYLOAD)
        self.trans = flat.array(nv.ARR_TRANS)
        self.sym_chars = flat.array(nv.ARR_SYMCHARS)
        self.start = flat.info[nv.INFO_START_STATE]
        self.wordhash = flat.array(nv.ARR_WORDHASH)
        self.sym_text_off = flat.array(nv.ARR_SYMTEXTOFF)
        self.sym_text = flat.array(nv.ARR_SYMTEXT)
        self.max_word_len = flat.max_word_len
        self.start_draw = flat.info[nv.INFO_START_DRAW] & 0xFFFFFFFF
        self.start_gbits = flat.info[nv.INFO_START_DRAW] >> 32
        self.len_in_rec = bool(flat.info[nv.INFO_MULTICHAR] & 2)    # (MC_DRAW_LEN_SHIFT: lengths in the draw records)

    @staticmethod
    def f64(q):
        return float(np.array([q[0], q[1]], dtype="<u4").view("<f8")[0])

    # the entry's DIRECT record for uniform r drawn in the row whose buckets start at goff
    def lookup(self, goff, gbits, r):
        K = 1 << gbits
        k = min(int(r * K), K - 1)
        q = self.draw[goff + k]
        typ = (int(q[3]) >> 29) & 3
        if typ != nv.DRAW_DIRECT:
            idx = int(q[2])
            if typ == nv.DRAW_ONE:
                idx += 1 if r >= self.f64(q) else 0
            else:
                lo, hi = 0, int(q[0])
                while lo < hi:
                    mid = (lo + hi) >> 1
                    if self.rcum[int(q[2]) + mid] > r:
                        hi = mid
                    else:
                        lo = mid + 1
                idx += lo
            return self.spay[idx], idx
        return q, None

    # mirrors k_sample: returns (ml2p, symbol ids, draws)
    def sample(self, uniform, min_len=4, max_len=256, keep_len=False, max_draws=1 << 16):
        goff, gbits, acc, plen, syms, draw = self.start_draw, self.start_gbits, 0.0, 0, [], 0
        while True:
            r = uniform()
            q, _ = self.lookup(goff, gbits, r)
            acc += self.f64(q)
            draw += 1
            restart = False
            meta = int(q[3])
            if meta & 0x80000000:
                if plen >= min_len:
                    return acc, syms, draw
                restart = True
            else:
                s = meta & 0xFFFFFF
                syms.append(s)
                if self.len_in_rec:
                    assert int(q[2]) >> 26 == int(self.sym_chars[s])
                    plen += int(q[2]) >> 26
                    goff = int(q[2]) & 0x03FFFFFF
                    goff = NONE if goff == 0x03FFFFFF else goff
                else:
                    plen += int(self.sym_chars[s])
                    goff = int(q[2])
                gbits = (meta >> 24) & 31
                if plen >= max_len:
                    restart = True
            if restart:
                goff, gbits, acc, syms = self.start_draw, self.start_gbits, 0.0, []
                if not keep_len:
                    plen = 0
            if draw >= max_draws or goff == NONE:
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

    def entry(self, state, s):
        row = state
        while row != NONE:
            e = self.find(row, s)
            if e != NONE:
                return e
            row = int(self.state[row]["parent"])
        return NONE

    def word(self, cps):
        h = 0xCBF29CE484222325
        for c in cps:
            h = ((h ^ c) * 0x100000001B3) & MASK64
        if h == MASK64:
            h -= 1
        mask = len(self.wordhash) - 1
        q = h & mask
        while True:
            sl = self.wordhash[q]
            if int(sl["key"]) == MASK64:
                return NONE
            if int(sl["key"]) == h and int(sl["pad"]) == len(cps):
                a = int(self.sym_text_off[int(sl["val"])])
                if list(self.sym_text[a:a + len(cps)]) == list(cps):
                    return int(sl["val"])
            q = (q + 1) & mask

    # mirrors k_score_words: (ml2p, split end offsets of the best parse)
    def score_words(self, text, end_text, iter_max, start_sym=NONE):
        cps = [ord(c) for c in text + end_text]
        total = len(cps)
        best, best_ends, done = 1022.0, [], 0
        stack = [[0.0, self.start, 0, 1]] if iter_max > 0 else []
        while stack:
            acc, state, pos, idx0 = stack[-1]
            top = min(total - pos, self.max_word_len)
            advanced = False
            for idx in range(idx0, top + 1):
                sym = self.word(cps[pos:pos + idx])
                if sym == NONE or sym == start_sym:
                    continue
                e = self.entry(state, sym)
                if e == NONE:
                    continue
                p = self.payload[e]
                nacc = acc + float(p["ml2p"])
                stack[-1][3] = idx + 1
                if pos + idx == total:
                    if best > nacc:
                        best, best_ends = nacc, [f[2] + f[3] - 1 for f in stack]
                    done += 1
                if done < iter_max:
                    stack.append([nacc, int(p["next"]), pos + idx, 1])
                advanced = True
                break
            if not advanced:
                stack.pop()
        return best, best_ends

    # mirrors k_sample_bpe: (ml2p, terminal entry indices, draws)
    def sample_bpe(self, uniform):
        seg_off, seg_row, _ = self.flat.seg
        q, _ = self.lookup(self.start_draw, self.start_gbits, uniform())
        acc, st, draws, ents = self.f64(q), int(q[2]), 1, []
        for k in range(int(seg_off[st]), int(seg_off[st + 1])):
            row = int(seg_row[k])
            if row == NONE:
                return float("nan"), ents, draws
            rec = self.state[row]
            q, _ = self.lookup(int(rec["guide_off"]), int(rec["gbits"]), uniform())
            draws += 1
            acc += self.f64(q)
            ents.append(int(q[2]))
        return acc, ents, draws

    # mirrors k_score_bpe
    def score_bpe(self, pwd, classes):
        seg_off, seg_row, seg_len = self.flat.seg
        units, j = [], 0
        while j < len(pwd):
            e = j + 1
            while e < len(pwd) and classes[e] == classes[j]:
                e += 1
            units.append((classes[j] << 24) | (e - j))
            j = e
        cand = self.flat.label_index.get(tuple(units), []) if pwd else []
        if not cand:
            cand = self.flat.np_index.get(len(pwd), [])
        best = 1022.0
        for st in cand:
            p, start, ok = float(self.payload[st]["ml2p"]), 0, True
            for k in range(int(seg_off[st]), int(seg_off[st + 1])):
                row, tl = int(seg_row[k]), int(seg_len[k])
                piece = pwd[start:start + tl]
                start += tl
                sym = self.word([ord(c) for c in piece]) if (row != NONE and piece) else NONE
                ent = self.find(row, sym) if sym != NONE else NONE
                if ent == NONE:
                    ok = False
                    break
                p += float(self.payload[ent]["ml2p"])
            if ok and p < 1022.0 and p < best:
                best = p
        return best


    # mirrors mc_enumerate (csrc/mc_enum.hip): the same expansion rule over state/payload, depth first
    def enumerate(self, threshold, min_len, max_len, backoff):
        out = []
        n_ctx = len(self.state)

        def expand(st, acc, ln, syms):
            if ln > max_len or acc >= threshold:
                return
            row = self.state[st]
            for j in range(int(row["len"])):
                p = self.payload[int(row["ent_off"]) + j]
                na = acc + float(p["ml2p"])
                if not na < threshold:
                    continue
                end, s = bool(int(p["sym"]) >> 31), int(p["sym"]) & 0x7FFFFFFF
                if end and (ln >= min_len or not backoff):
                    if ln >= min_len:
                        out.append((list(syms), na))
                    continue
                nxt = int(p["next"])
                assert nxt < n_ctx, "context without a row"
                expand(nxt, na, ln + int(self.sym_chars[s]), syms + [s])

        expand(self.start, 0.0, 0, [])
        return out
