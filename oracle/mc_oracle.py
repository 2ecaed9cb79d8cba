"""CPU restatement of kofny/MonteCarlo's Monte Carlo guess-number path.

TEST INFRASTRUCTURE ONLY.  Nothing under montecarlo_amd/ may import this file; it
is the checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg).  It restates, in the reference's own data terms (dict rows
keyed by tuple contexts, numpy fp64 cumulative sums), the algorithms of

  lib4mc/ProbLib.py:10-43            expand_1d / expand_2d / pick_expand
  lib4mc/MonteCarloParent.py:33-69   sample / parse_file
  lib4mc/MonteCarloLib.py:10-68      rank estimator
  nwords/nwords_trainer.py:22-53     fixed-order counter
  backwords/backwords_trainer.py:23-78            backoff counter
  backwords/backwords_secondary_trainer.py:23-121 count merge + freq2prob
  nwords_v2/nwords_trainer_v2.py:12-56            v2 counter
  nwords/nwords_enumerator.py:15-42, backwords/backwords_enumerator.py:15-45   threshold enumerators
  nwords_simulator.py:30-116, backwords_simulator.py:28-49,
  nwords_simulator_v2.py:60-82, bpe_simulator.py:17-150, bpeX/modelreader.py:15-68

Pinning: tests/golden/make_golden.py imports the reference itself in the build
container and records its outputs; tests/test_oracle_golden.py checks every
function here against those fixtures bit for bit (fp64 as hex).

Deliberate differences from the reference (each recorded in the fixtures):
  * uniforms come from a callable, not from ``random.uniform``;
  * ``pick`` clamps to the last entry when ``u == cum[-1]`` where the reference
    raises IndexError (ProbLib.py:41-42);
  * float sums are explicit left-to-right loops (CPython 3.12 changed ``sum``).
"""
from __future__ import annotations

import bisect
import math
import os
import re
from typing import Callable, Dict, Iterable, List, Sequence, Tuple

import numpy as np

UNPARSEABLE = float(1022)          # nwords_simulator.py:85 ; = -log2(sys.float_info.min)
MAX_PWD_LEN = 256                  # nwords_simulator.py:112


# --------------------------------------------------------------------------- tables
def expand_row(row: Dict, minus_log_based: bool = False):
    """ProbLib.py:20-28: (value dict, key list, sequential fp64 cumsum of raw p)."""
    keys = list(row)
    cum = np.cumsum(np.array([row[k] for k in keys], dtype=np.float64))
    vals = {k: -math.log2(v) for k, v in row.items()} if minus_log_based else row
    return vals, keys, cum


def expand_table(table: Dict, minus_log_based: bool = False):
    """ProbLib.py:10-17: rows with no entries are dropped."""
    return {ctx: expand_row(row, minus_log_based) for ctx, row in table.items() if len(row)}


def pick(expanded, uniform: Callable[[], float]):
    """ProbLib.py:31-43.  random.uniform(0, t) is 0 + (t - 0) * random()."""
    vals, keys, cum = expanded
    total = cum[-1]
    u = 0.0 + (total - 0.0) * uniform()
    idx = bisect.bisect_right(cum, u)
    if idx >= len(keys):           # reference: IndexError; build: clamp (documented)
        idx = len(keys) - 1
    k = keys[idx]
    return vals[k], k


# --------------------------------------------------------------------------- trainers
def _split(line: str, splitter: str, start: int, step: int) -> List[str]:
    """nwords_trainer.py:13-19 (the splitter is a regular expression)."""
    line = line.strip("\r\n")
    if splitter == "":
        return list(line)
    items = re.split(splitter, line)
    return items[start:len(items):step]


def train_nwords(lines: Iterable[str], n: int, splitter: str, end_chr: str, start4word: int = 0,
                 skip4word: int = 1, start_chr: str = "\x00"):
    """nwords_trainer.py:22-53.  Row/key order = first appearance while scanning the
    unique section tuples in first-seen order."""
    order = n - 1
    uniq: Dict[Tuple, int] = {}
    words: Dict[str, int] = {}
    for line in lines:
        secs = [start_chr] * order + _split(line, splitter, start4word, skip4word) + [end_chr]
        for s in secs:
            words[s] = words.get(s, 0) + 1
        t = tuple(secs)
        uniq[t] = uniq.get(t, 0) + 1
    counts: Dict[Tuple, Dict[str, int]] = {}
    for secs, c in uniq.items():
        for i in range(len(secs) - order):
            row = counts.setdefault(secs[i:i + order], {})
            tr = secs[i + order]
            row[tr] = row.get(tr, 0) + c
    probs = {}
    for ctx, row in counts.items():
        tot = 0
        for v in row.values():
            tot += v
        probs[ctx] = {k: v / tot for k, v in row.items()}
    return probs, words


def train_backwords(lines: Iterable[str], splitter: str, start_chr: str, end_chr: str, start4word: int,
                    skip4word: int, threshold: int, max_gram: int):
    """backwords_trainer.py:23-78."""
    zero_cnt: Dict[str, int] = {}
    words: Dict[str, int] = {}
    by_len: Dict[int, Dict[Tuple, int]] = {}
    for line in lines:
        secs = [start_chr] + _split(line, splitter, start4word, skip4word) + [end_chr]
        for s in secs:
            words[s] = words.get(s, 0) + 1
            if s != start_chr:
                zero_cnt[s] = zero_cnt.get(s, 0) + 1
        grp = by_len.setdefault(len(secs), {})
        t = tuple(secs)
        grp[t] = grp.get(t, 0) + 1
    ztot = 0
    for v in zero_cnt.values():
        ztot += v
    probs: Dict[Tuple, Dict[str, float]] = {(): {k: v / ztot for k, v in zero_cnt.items()}}
    lens_ok = [ln for ln, grp in by_len.items() if sum(grp.values()) >= threshold]
    top = min(max(lens_ok), max(2, max_gram))
    for n in range(2, top + 1):
        order = n - 1
        counts: Dict[Tuple, Dict[str, int]] = {}
        for ln, grp in by_len.items():
            if n > ln:
                continue
            for secs, c in grp.items():
                for i in range(len(secs) - order):
                    row = counts.setdefault(secs[i:i + order], {})
                    tr = secs[i + order]
                    row[tr] = row.get(tr, 0) + c
        for ctx, row in counts.items():
            tot = 0
            for v in row.values():
                tot += v
            if tot < threshold:
                continue
            kept = {k: v / tot for k, v in row.items() if v >= threshold}
            acc = 0                 # Python-3.10 sum(): int 0 then left-to-right
            for v in kept.values():
                acc = acc + v
            missing = 1 - acc
            if missing == 1:
                continue
            if missing > 0:
                for k, p in probs[ctx[1:]].items():
                    kept[k] = kept.get(k, 0) + p * missing
            probs[ctx] = kept
    return probs, words


def merge_secondary_counts(lines, splitter: str, start_chr: str, end_chr: str, start4word: int, skip4word: int,
                           max_gram: int, threshold: int, counts=None, words=None):
    """backwords_secondary_trainer.py:23-99 (raw counts merged across rounds)."""
    if counts is None:
        counts, words = {}, {}
    lines = list(lines) if not isinstance(lines, list) else lines
    if isinstance(lines, list) and len(lines) == 0:
        return counts, words
    by_len: Dict[int, Dict[Tuple, int]] = {}
    longest = 2
    for line in lines:
        secs = [start_chr] + _split(line, splitter, start4word, skip4word) + [end_chr]
        for s in secs:
            words[s] = words.get(s, 0) + 1
            if s != start_chr:
                z = counts.setdefault((), {})
                z[s] = z.get(s, 0) + 1
        grp = by_len.setdefault(len(secs), {})
        t = tuple(secs)
        grp[t] = grp.get(t, 0) + 1
        longest = max(longest, len(secs))
    for n in range(2, min(max_gram, longest) + 1):
        order = n - 1
        fresh: Dict[Tuple, Dict[str, int]] = {}
        for ln, grp in by_len.items():
            if ln < n:
                continue
            for secs, c in grp.items():
                for i in range(ln - order):
                    row = fresh.setdefault(secs[i:i + order], {})
                    tr = secs[i + order]
                    row[tr] = row.get(tr, 0) + c
        if not fresh:
            break
        for ctx, row in fresh.items():
            if ctx not in counts:
                if any(c >= threshold for c in row.values()):
                    counts[ctx] = row
                continue
            have = counts[ctx]
            for tr, c in row.items():
                have[tr] = have.get(tr, 0) + c
    return counts, words


def freq2prob(counts: Dict[Tuple, Dict[str, int]], threshold: int):
    """backwords_secondary_trainer.py:102-121: shortest contexts first (stable),
    zero-gram thresholded and NOT renormalised."""
    probs: Dict[Tuple, Dict[str, float]] = {}
    for ctx, row in sorted(counts.items(), key=lambda kv: len(kv[0])):
        tot = 0
        for v in row.values():
            tot += v
        kept = {k: v / tot for k, v in row.items() if v >= threshold}
        if not kept:
            continue
        if len(kept) < len(row) and len(ctx) > 0:
            acc = 0
            for v in kept.values():
                acc = acc + v
            missing = 1.0 - acc
            for k, p in probs[ctx[1:]].items():
                kept[k] = kept.get(k, .0) + p * missing
        probs[ctx] = kept
    return probs


def train_v2(lines: Iterable[str], n: int = 4, end_chr: str = "\x03", threshold: int = 10):
    """nwords_trainer_v2.py:12-56: contexts are the last n-1 CHARACTERS, transitions
    are words (rare words exploded into characters)."""
    order = n - 1
    words: Dict[str, int] = {}
    uniq: Dict[Tuple, int] = {}
    for line in lines:
        items = line.strip("\r\n").split("\t")
        pwd = items[0] + end_chr
        pos, secs = 0, []
        for raw in items[1::2] + [end_chr]:
            w = pwd[pos:pos + len(raw)]
            secs.append(w)
            pos += len(raw)
            words[w] = words.get(w, 0) + 1
        if "".join(secs) != pwd or len(pwd) < 4:
            raise Exception("error1")
        t = tuple(secs)
        uniq[t] = uniq.get(t, 0) + 1
    keep = {w for w, c in words.items() if c >= threshold}
    counts: Dict[str, Dict[str, int]] = {}
    for secs, c in uniq.items():
        flat: List[str] = []
        for s in secs:
            if s in keep:
                flat.append(s)
            else:
                flat.extend(s)
        ctx = ""
        for s in flat:
            row = counts.setdefault(ctx, {})
            row[s] = row.get(s, 0) + c
            ctx = (ctx + s)[-order:]
    probs = {}
    for ctx, row in counts.items():
        tot = 0
        for v in row.values():
            tot += v
        probs[ctx] = {k: v / tot for k, v in row.items()}
    return probs, words


# --------------------------------------------------------------------------- models
class NGramOracle:
    """Fixed-order model: nwords_simulator.py:13-116."""

    def __init__(self, probs, words, n: int, start_chr: str = "\x00", end_chr: str = "\x03"):
        self.table = expand_table(probs)
        self.words = words
        self.n = n
        self.start_chr, self.end_chr = start_chr, end_chr
        self.min_len = 4
        self.start_hist = tuple([start_chr] * (n - 1))
        self.max_iter = 10 ** 35

    # nwords_simulator.py:30-34
    def context(self, hist: Sequence[str], transition: str) -> Tuple:
        return tuple(hist) if len(hist) < self.n else tuple(hist[1 - self.n:])

    # nwords_simulator.py:92-116
    def sample1(self, uniform: Callable[[], float]):
        hist, acc, plen = list(self.start_hist), .0, 0
        while True:
            p, sym = pick(self.table[self.context(hist, "")], uniform)
            acc += -math.log2(p)
            if sym == self.end_chr:
                if plen >= self.min_len:
                    break
                hist, acc, plen = list(self.start_hist), .0, 0
                continue
            hist.append(sym)
            plen += len(sym)
            if plen >= MAX_PWD_LEN:
                hist, acc, plen = list(self.start_hist), .0, 0
        return acc, "".join(s for s in hist if s != self.start_chr)

    # nwords_simulator.py:36-60,84-90 / backwords_simulator.py:43-49, as an explicit stack
    def calc_ml2p(self, pwd: str):
        text = pwd + self.end_chr
        target = len(pwd) + len(self.end_chr)
        best, best_parts = UNPARSEABLE, [pwd]
        hist: List[str] = list(self.start_hist)
        probs: List[float] = []
        done = 0
        # frame = [offset into text, next split index to try, pushed?]
        stack = [[0, 1, False]] if done < self.max_iter else []
        while stack:
            fr = stack[-1]
            off, idx = fr[0], fr[1]
            if fr[2]:                       # returning from the child of split idx-1
                hist.pop()
                probs.pop()
                fr[2] = False
            rest = len(text) - off
            advanced = False
            while idx <= rest:
                left = text[off:off + idx]
                idx += 1
                if left not in self.words:
                    continue
                ctx = self.context(hist, left)
                row = self.table.get(ctx)
                if row is None or left not in row[0]:
                    continue
                hist.append(left)
                probs.append(row[0][left])
                parts = [c for c in hist if c != self.start_chr]
                if len("".join(parts)) == target:
                    tot = 0
                    for p in probs:
                        tot = tot + (-math.log2(p))
                    if best > tot:
                        best, best_parts = tot, list(parts)
                    done += 1
                fr[1], fr[2] = idx, True
                if done < self.max_iter:        # callee's entry check (:39)
                    stack.append([off + len(left), 1, False])
                advanced = True
                break
            if not advanced:
                stack.pop()
        return best, [c for c in best_parts if c != self.end_chr]


class BackoffOracle(NGramOracle):
    """backwords_simulator.py:11-49 (also the secondary model, :11-25 of
    backwords_secondary_simulator.py, which only swaps the table source)."""

    def __init__(self, probs, words, start_chr: str = "\x00", end_chr: str = "\x03", max_iter: int = 10 ** 100):
        self.table = expand_table(probs)
        self.words = words
        self.n = None
        self.start_chr, self.end_chr = start_chr, end_chr
        self.min_len = 4
        self.start_hist = tuple(start_chr)          # list(default_start) of a str
        self.max_iter = max_iter

    # backwords_simulator.py:28-41
    def context(self, hist: Sequence[str], transition: str) -> Tuple:
        for i in range(len(hist)):
            ctx = tuple(hist[i:])
            row = self.table.get(ctx)
            if row is None:
                continue
            if transition != "" and transition not in row[0]:
                continue
            return ctx
        return ()


class V2Oracle:
    """nwords_simulator_v2.py:60-82 -- sampling only (its calc_ml2p is defective,
    SURVEY.md §3.4).  pwd_len is NOT reset on restart, as in the reference."""

    def __init__(self, probs, words, n: int, end_chr: str = "\x03"):
        self.table = expand_table(probs)
        self.words, self.n, self.end_chr = words, n, end_chr

    def sample1(self, uniform: Callable[[], float], max_draws: int = 1 << 20):
        pwd, acc, plen, draws = "", .0, 0, 0
        while True:
            if draws >= max_draws:
                raise RuntimeError("v2 sampler did not terminate (reference would hang)")
            ctx = pwd if plen < self.n else pwd[1 - self.n:]
            p, sym = pick(self.table[ctx], uniform)
            draws += 1
            acc += -math.log2(p)
            if sym == self.end_chr:
                if plen > 3:
                    break
                pwd, acc = "", .0
                continue
            pwd += sym
            plen += len(sym)
            if plen >= MAX_PWD_LEN:
                pwd, acc = "", .0
        return acc, pwd


# --------------------------------------------------------------------------- BPE-PCFG
def luds(pwd: str):
    """bpe_simulator.py:17-40."""
    out, prev, run, cur = [], "", 0, " "
    for ch in pwd:
        if ch.isalpha():
            cur = "U" if ch.isupper() else "L"
        elif ch.isdigit():
            cur = "D"
        else:
            cur = "S"
        if cur == prev:
            run += 1
        else:
            if prev:
                out.append((prev, run))
            prev, run = cur, 1
    out.append((cur, run))
    return tuple(out)


def read_bpe(model_path: str):
    """bpeX/modelreader.py:15-68."""
    def read_tag(sub, tag):
        out: Dict[Tuple[str, int], Dict[str, float]] = {}
        for root, _dirs, files in os.walk(os.path.join(model_path, sub)):
            for fn in files:
                key = (tag, int(fn[:fn.find(".")]))
                with open(os.path.join(root, fn)) as fd:
                    for line in fd:
                        t, p = line.strip("\r\n").split("\t")
                        out.setdefault(key, {})[t] = float(p)
        return out

    grammars = {}
    piece = re.compile(r"([A-Z]+)([0-9]+)")
    with open(os.path.join(model_path, "grammar", "structures.txt")) as fd:
        for line in fd:
            raw, p = line.strip("\r\n").split("\t")
            grammars[tuple((m.group(1), int(m.group(2))) for m in piece.finditer(raw))] = float(p)
    terminals = {}
    for sub, tag in (("lower", "L"), ("upper", "U"), ("mixed_2", "DM"), ("mixed_3", "TM"), ("mixed_4", "FM"),
                     ("digits", "D"), ("special", "S")):
        terminals.update(read_tag(sub, tag))
    return grammars, terminals


class BpeOracle:
    """bpe_simulator.py:45-157."""

    def __init__(self, grammars, terminals):
        self.grammars = expand_row(grammars, minus_log_based=True)
        self.terminals = expand_table(terminals, minus_log_based=True)
        self.converted, self.not_parsed = self._count_luds(grammars)

    @staticmethod
    def _count_luds(structures):
        # bpe_simulator.py:45-101
        skipped, conv = [], {}
        for st in structures:
            merged, has_m = [], False
            for tag, ln in st:
                if merged and merged[-1][0] == tag:
                    merged[-1] = (tag, merged[-1][1] + ln)
                else:
                    merged.append((tag, ln))
                has_m = has_m or ("M" in tag)
            if has_m:
                skipped.append(st)
            else:
                conv.setdefault(tuple(merged), set()).add(st)
        by_total: Dict[int, set] = {}
        for k in conv:
            by_total.setdefault(sum(ln for _, ln in k), set()).add(k)
        not_parsed: Dict[int, set] = {}
        for st in skipped:
            total = sum(ln for _, ln in st)
            flat_s = [t for t, ln in st for _ in range(ln)]
            for cand in by_total.get(total, ()):
                flat_c = [t for t, ln in cand for _ in range(ln)]
                if len(flat_c) == len(flat_s) and all(
                        a == b or "M" in a or "M" in b for a, b in zip(flat_c, flat_s)):
                    conv[cand].add(st)
            not_parsed.setdefault(total, set()).add(st)
        return conv, not_parsed

    def sample1(self, uniform: Callable[[], float]):
        acc, pwd = .0, ""
        p, struct = pick(self.grammars, uniform)
        acc += p
        for seg in struct:
            p, term = pick(self.terminals[seg], uniform)
            acc += p
            pwd += term
        return acc, pwd

    def calc_ml2p(self, pwd: str) -> float:
        label = luds(pwd)
        cands = set(self.converted.get(label, ()))
        if not cands:
            cands |= self.not_parsed.get(sum(ln for _, ln in label), set())
            if not cands:
                return UNPARSEABLE
        gram = self.grammars[0]
        best = UNPARSEABLE
        for cand in cands:
            p = gram.get(cand, UNPARSEABLE)
            if p == UNPARSEABLE:
                break
            pos = 0
            for seg in cand:
                term = self.terminals.get(seg)[0]
                piece = pwd[pos:pos + seg[1]]
                pos += seg[1]
                if piece not in term:
                    p = UNPARSEABLE
                    break
                p += term[piece]
            if p < best:
                best = p
        return best


# --------------------------------------------------------------------------- driver-level pieces
def sample_many(model, size: int, uniform_for: Callable[[int], Callable[[], float]]):
    """MonteCarloParent.py:33-46: draw-order list of ml2p and the first-seen-ordered
    dict  pwd -> [ml2p, count]."""
    out, seen = [], {}
    for i in range(size):
        ml2p, pwd = model.sample1(uniform_for(i))
        out.append(ml2p)
        if pwd not in seen:
            seen[pwd] = [ml2p, 0]
        seen[pwd][1] += 1
    return out, seen


def parse_lines(model, lines: Iterable[str], using_component: bool = False):
    """MonteCarloParent.py:48-69: dedup, score unique lines, stable sort by ml2p."""
    cnt: Dict[str, int] = {}
    for line in lines:
        line = line.strip("\r\n")
        cnt[line] = cnt.get(line, 0) + 1
    res = []
    for pwd, c in cnt.items():
        ml2p, parts = model.calc_ml2p(pwd)
        res.append((parts if using_component else pwd, c, ml2p))
    res.sort(key=lambda t: t[2])
    return res


class RankOracle:
    """lib4mc/MonteCarloLib.py:9-68."""

    def __init__(self, ml2p_list: List[float]):
        ml2p_list.sort()                                   # in place, :11
        self.keys = np.fromiter(ml2p_list, float)
        self.positions = (2 ** (self.keys - math.log2(len(self.keys)))).cumsum()

    def ml2p2rank(self, x):
        idx = bisect.bisect_right(self.keys, x)
        return self.positions[idx - 1] if idx > 0 else 1

    def iter2gc(self, scored, need_resort: bool = False, add1: bool = True):
        if need_resort:
            scored = sorted(scored, key=lambda t: t[2])
        total = 0
        for _, c, _ in scored:
            total += c
        step = 1 if add1 else 0
        out, prev, cracked = [], 0, 0
        for pwd, c, x in scored:
            rank = math.ceil(max(self.ml2p2rank(x), prev + step))
            cracked += c
            prev = rank
            out.append((pwd, x, c, rank, cracked, cracked / total * 100))
        return out

    @staticmethod
    def format_gc(gc) -> str:
        return "".join(f"{pwd}\t{x:.8f}\t{c}\t{rank}\t{cracked}\t{ratio:5.2f}\n"
                       for pwd, x, c, rank, cracked, ratio in gc)


# --------------------------------------------------------------------------- threshold enumerators
def enumerate_strings(table: Dict, threshold: float, start_chr: str, end_chr: str, min_len: int, order: int = None):
    """nwords/nwords_enumerator.py:15-42 (``order`` given: context = last ``order`` symbols, a node with
    cur_len > 30 is not expanded, an end symbol never recurses) and backwords/backwords_enumerator.py:15-45
    (``order`` None: context = longest suffix that is a row, cur_len > 256, an end symbol met before
    min_len is appended and the walk goes on).  ``table`` holds probabilities; returns the list of
    (string, -log2 p) in the order the reference writes them.  The recursion is an explicit stack of
    (history, sum, length, iterator over the row)."""
    ml2p = {ctx: [(s, -math.log2(p)) for s, p in row.items()] for ctx, row in table.items()}
    backoff = order is None
    limit = 256 if backoff else 30

    def row_of(hist: Tuple):
        if not backoff:
            return ml2p[hist[-order:]]                    # KeyError as in the reference (:26)
        for i in range(len(hist) + 1):
            if hist[i:] in ml2p:
                return ml2p[hist[i:]]
        raise KeyError(hist)

    out: List[Tuple[str, float]] = []
    lead = 1 if backoff else order
    root = (start_chr,) * lead
    stack = []

    def enter(hist, acc, length):
        row = row_of(hist)
        if length > limit or acc >= threshold:
            return
        stack.append((hist, acc, length, iter(row)))

    enter(root, 0.0, 0)
    while stack:
        hist, acc, length, it = stack[-1]
        step = next(it, None)
        if step is None:
            stack.pop()
            continue
        sym, m = step
        new = acc + m
        if not new < threshold:
            continue
        if sym == end_chr and (length >= min_len or not backoff):
            if length >= min_len:
                out.append(("".join(hist[lead:]), new))
            continue
        enter(hist + (sym,), new, length + len(sym))
    return out
