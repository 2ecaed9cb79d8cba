"""CPU tests of the host side of the product: trainers reproduce the reference's
tables (golden digests), the C-ABI library loads and exports every declared
symbol, and the compiled walk automaton -- stepped by tests/walker.py exactly as
the HIP kernels step it -- reproduces the reference's samples and scores."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, fh, golden, golden_lines, hx
from montecarlo_amd import _native as nv
from montecarlo_amd import synth
from montecarlo_amd.backwords.backwords_secondary_trainer import backwords_counter as secondary_counter
from montecarlo_amd.backwords_secondary_simulator import BackWordsSecondaryMonteCarlo
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.nwords_simulator import NWordsMonteCarlo
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
from oracle.philox import SampleStream
from walker import CompiledWalker


def rows_of(table):
    return [[ctx if isinstance(ctx, str) else list(ctx), list(keys), [hx(vals[k]) for k in keys],
             [hx(c) for c in cum]] for ctx, (vals, keys, cum) in table.items()]


def digest(obj):
    return hashlib.sha256(json.dumps(obj, ensure_ascii=True, separators=(",", ":")).encode()).hexdigest()


def test_library_exports_header_symbols():
    hdr = open(os.path.join(ROOT, "include", "mi355mc.h")).read()
    declared = set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", hdr)) - {"mc_table_info_idx"}
    assert declared == set(nv.EXPORTS), declared ^ set(nv.EXPORTS)
    lib = ctypes.CDLL(nv.LIB_PATH)
    for name in declared:
        assert getattr(lib, name) is not None
    assert nv.lib.mc_abi_version() == 2


def check_table(fx, model):
    rows = rows_of(model.nwords)
    assert len(rows) == fx["n_ctx"]
    assert rows[:len(fx["rows_head"])] == fx["rows_head"]
    assert digest(rows) == fx["table_digest"]
    if "words_digest" in fx:
        assert digest(sorted(model.words.items())) == fx["words_digest"]
    assert model._flat.info[nv.INFO_NCTX] == fx["n_ctx"]
    assert model._flat.info[nv.INFO_NENT] == fx["n_entries"]


def check_walk(fx, model, n_samples=150, n_scores=None, keep_len=False):
    flat = model._flat
    w = CompiledWalker(flat)
    seed = fx["samples"]["seed"]
    drop = getattr(model, "start_chr", "")
    for i, (want, pwd, draws) in enumerate(fx["samples"]["items"][:n_samples]):
        st = SampleStream(seed, i)
        acc, syms, used = w.sample(st, keep_len=keep_len)
        got = "".join(c for c in (flat.symbols[s] for s in syms) if c != drop)
        assert (hx(acc), got, used) == (want, pwd, draws), i
    if "scores" in fx and not flat.multichar:
        for pwd, want, _parts in fx["scores"][:n_scores]:
            ids, _ = flat.encode_chars([pwd])
            got = w.score(ids.tolist(), flat.end_sym, len(flat.symbols))
            assert hx(got) == want, repr(pwd)
    if "scores" in fx:
        check_dfs(fx, model, w, 60 if not flat.multichar else n_scores)


def check_dfs(fx, model, w, n_scores=None, iter_max=None):
    """The DFS walk (k_score_words' mirror) against the reference's calc_ml2p and components."""
    end = model.end_chr
    iter_max = iter_max if iter_max is not None else getattr(model, "max_iter", 10 ** 35)
    for pwd, want, parts in fx["scores"][:n_scores]:
        got, ends = w.score_words(pwd, end, iter_max)
        assert hx(got) == want, repr(pwd)
        text, toks, prev = pwd + end, [], 0
        for e in ends:
            toks.append(text[prev:e])
            prev = e
        toks = [c for c in toks if c != end] if got < 1022.0 else ([pwd] if pwd != end else [])
        assert toks == parts, repr(pwd)


@pytest.mark.parametrize("n", [2, 4])
def test_nwords_char(train_char, n):
    fx = golden(f"nwords_n{n}_char.json")
    m = NWordsMonteCarlo(synth.as_file(train_char), n=n, splitter="")
    check_table(fx, m)
    check_walk(fx, m)


def test_nwords_tab(train_tab):
    fx = golden("nwords_n3_tab.json")
    m = NWordsMonteCarlo(synth.as_file(train_tab), n=3, splitter="\t")
    check_table(fx, m)
    assert m._flat.multichar and m._flat.len_in_rec       # (the draw records carry the symbols' lengths: walker.sample reads them)
    check_walk(fx, m)


@pytest.mark.parametrize("thr", [2, 10])
def test_backwords_char(train_char, thr):
    fx = golden(f"backwords_thr{thr}_char.json")
    m = BackWordsMonteCarlo(synth.as_file(train_char), splitter="", threshold=thr)
    check_table(fx, m)
    check_walk(fx, m, n_samples=250)


def test_backwords_tab(train_tab):
    fx = golden("backwords_thr3_tab.json")
    m = BackWordsMonteCarlo(synth.as_file(train_tab), splitter="\t", threshold=3)
    check_table(fx, m)
    check_walk(fx, m)
    # --max-iter 2: the parse cap of the recursion (entry check only, callers finish their loops)
    check_dfs(golden("backwords_thr3_tab_maxiter2.json"), m, CompiledWalker(m._flat), iter_max=2)


def test_secondary(train_char):
    fx = golden("secondary_thr4_char.json")
    cfg = {"start_chr": "\x03", "end_chr": "\x00", "threshold": 4}
    counts, words = secondary_counter(train_char[:2500], "", "\x03", "\x00", 0, 1, 256, 4)
    m = BackWordsSecondaryMonteCarlo((counts, words, cfg))
    check_table(fx["round0"], m)
    check_walk(fx["round0"], m, n_samples=100)
    counts, words = secondary_counter(train_char[2500:3200], "", "\x03", "\x00", 0, 1, 256, 4, nwords_dict=counts,
                                      words=words)
    assert digest([[list(k), sorted(v.items())] for k, v in counts.items()]) == fx["meta"]["counts_digest"]
    m = BackWordsSecondaryMonteCarlo((counts, words, cfg))
    check_table(fx["round1"], m)
    check_walk(fx["round1"], m, n_samples=100)


def test_v2(train_v2):
    """The native v2 trainer (mc_train_v2) and the dict-based one build the reference's table (golden digest of
    rows in dict order, fp64 as hex) and the reference's `words`; the compiled walk reproduces its samples."""
    from montecarlo_amd.lib4mc.ProbLib import expand_2d
    from montecarlo_amd.nwords_v2.nwords_trainer_v2 import nwords_counter as v2_counter
    fx = golden("v2_n4.json")
    m = NWords2MonteCarlo(synth.as_file(train_v2), 4)
    assert hasattr(m, "_trained")
    check_table(fx, m)
    check_walk(fx, m, keep_len=True)
    probs, words = v2_counter(synth.as_file(train_v2), 4, "\x03")
    assert rows_of(expand_2d(probs)) == rows_of(m.nwords)
    assert list(words.items()) == list(m.words.items())
    # other orders, a larger synthetic corpus, segments that run past the password (accepted by the reference:
    # its slices clamp), and the reference's "error1" when they fall short
    for n, lines in ((6, synth.v2_corpus(3000)), (3, train_v2[:500] + ["abcdef\tabc\tL\tdefgh\tL"])):
        nat = NWords2MonteCarlo(synth.as_file(lines), n)
        probs, words = v2_counter(synth.as_file(lines), n, "\x03")
        assert rows_of(expand_2d(probs)) == rows_of(nat.nwords)
        assert list(words.items()) == list(nat.words.items())
    with pytest.raises(nv.NativeError, match="error1"):
        NWords2MonteCarlo(synth.as_file(["abcdef\tabc\tL\tde\tL"]), 4)
    with pytest.raises(Exception, match="error1"):
        v2_counter(synth.as_file(["abcdef\tabc\tL\tde\tL"]), 4, "\x03")


def test_bpe():
    """BPE-PCFG (bpe_simulator.py:104-157): the two-level compiled table walked as k_sample_bpe /
    k_score_bpe walk it, against the reference's samples and calc_ml2p values (golden/bpe.json)."""
    from montecarlo_amd.bpe_simulator import BpePcfgSim, char_class, luds
    fx = golden("bpe.json")
    m = BpePcfgSim(os.path.join(GOLDEN, "bpe_model"))
    w = CompiledWalker(m._flat)
    seed = fx["samples"]["seed"]
    for i, (want, pwd, draws) in enumerate(fx["samples"]["items"][:300]):
        acc, ents, used = w.sample_bpe(SampleStream(seed, i))
        assert (hx(acc), "".join(m._flat.entry_text[e] for e in ents), used) == (want, pwd, draws), i
    for pwd, want in fx["scores"]:
        assert hx(w.score_bpe(pwd, [char_class(c) for c in pwd])) == want, pwd
    assert luds("") == ((" ", 0),) and luds("ab1!") == (("L", 2), ("D", 1), ("S", 1))


def test_draw_table_equals_bisect(train_char):
    """The bucketed draw table returns bisect_right(cum, total*r) (clamped to the last entry, the
    reference raises there) for r at every bucket's edges, just inside them, at every threshold
    and its two neighbours, and at random r -- and every DIRECT record equals its entry's."""
    m = BackWordsMonteCarlo(synth.as_file(train_char), splitter="", threshold=10)
    w = CompiledWalker(m._flat)
    rng = np.random.default_rng(5)
    kinds = set()
    for r_ in rng.choice(len(w.state), size=300, replace=False):
        rec = w.state[r_]
        e0, L, g = int(rec["ent_off"]), int(rec["len"]), int(rec["gbits"])
        K, goff, total = 1 << g, int(rec["guide_off"]), float(rec["total"])
        cum = w.cum[e0:e0 + L]
        probes = []
        for k in range(K):
            probes += [k / K, np.nextafter((k + 1) / K, 0.0), (k + 0.5) / K]
            kinds.add((int(w.draw[goff + k][3]) >> 29) & 3)
        for t in w.rcum[e0:e0 + L - 1]:
            probes += [float(t), float(np.nextafter(t, 0.0)), float(np.nextafter(t, 2.0))]
        probes += rng.random(50).tolist()
        for r in probes:
            if not 0.0 <= r < 1.0:
                continue
            want = min(int(np.searchsorted(cum, total * r, side="right")), L - 1)
            q, idx = w.lookup(goff, g, r)
            assert np.array_equal(q, w.spay[e0 + want]), (r_, r)
            assert idx is None or idx == e0 + want
    assert kinds == {nv.DRAW_DIRECT, nv.DRAW_ONE, nv.DRAW_MULTI}


def test_dense_transitions_equal_backoff_chain(train_char):
    """The dense (row, symbol) table k_score<DENSE> gathers from holds exactly what walking the
    parent chain with the transition hash finds (longest suffix row that lists the symbol)."""
    for m in (BackWordsMonteCarlo(synth.as_file(train_char), splitter="", threshold=10),
              NWordsMonteCarlo(synth.as_file(train_char), n=4, splitter="")):
        w = CompiledWalker(m._flat)
        dense = m._flat.array(nv.ARR_DENSE)
        n_sym = len(m._flat.symbols)
        assert len(dense) == len(w.state) * n_sym
        rng = np.random.default_rng(9)
        for row in rng.choice(len(w.state), size=150, replace=False):
            for s in range(n_sym):
                e = w.entry(int(row), s)
                got = dense[int(row) * n_sym + s]
                assert int(got["sym"]) == e
                if e != nv.MC_NONE:
                    assert (got["ml2p"], got["next"]) == (w.payload[e]["ml2p"], w.payload[e]["next"])
    sub = NWordsMonteCarlo(synth.as_file(golden_lines("train_tab.txt")), n=3, splitter="\t")
    assert len(sub._flat.array(nv.ARR_DENSE)) == 0          # sub-word alphabets keep the hash + DFS path


def test_native_trainers_equal_python_trainers(train_char, train_tab):
    """csrc/mc_train.cpp against the Python trainers of the same contract (which serve the
    regular-expression splitters): identical rows, orders, probabilities and `words`."""
    from montecarlo_amd.backwords.backwords_trainer import backwords_counter
    from montecarlo_amd.lib4mc.ProbLib import expand_2d
    from montecarlo_amd.nwords.nwords_trainer import nwords_counter
    for lines, splitter, thr in ((train_char, "", 10), (train_tab, "\t", 3), (train_char + ["", "x", "\r"], "", 2)):
        m = BackWordsMonteCarlo(synth.as_file(lines), splitter=splitter, threshold=thr)
        assert hasattr(m, "_trained")
        probs, words = backwords_counter(synth.as_file(lines), splitter, "\x00", "\x03", 0, 1, thr, 256)
        assert rows_of(m.nwords) == rows_of(expand_2d(probs))
        assert list(m.words.items()) == list(words.items())
    for lines, splitter, n in ((train_char, "", 3), (train_tab, "\t", 2)):
        m = NWordsMonteCarlo(synth.as_file(lines), n=n, splitter=splitter)
        assert hasattr(m, "_trained")
        probs, words = nwords_counter(synth.as_file(lines), n, splitter, "\x03", 0, 1)
        assert rows_of(m.nwords) == rows_of(expand_2d(probs))
        assert list(m.words.items()) == list(words.items())
    # a regular-expression splitter is served by the Python trainer
    m = NWordsMonteCarlo(synth.as_file(train_tab), n=2, splitter="\t+")
    assert not hasattr(m, "_trained") and len(m.nwords) > 0


def test_native_secondary_counts_equal_python(train_char):
    """mc_counts (merge + freq2prob in C++) against the Python secondary trainer: same raw counts (content
    and dict order) after every merge, same probability tables, and the pickle is a plain dict."""
    import pickle
    from montecarlo_amd.backwords.backwords_secondary_trainer import freq2prob
    from montecarlo_amd.lib4mc.ProbLib import expand_2d
    from montecarlo_amd.trained import NativeCounts
    import montecarlo_amd.backwords.backwords_secondary_trainer as sec_mod
    rounds = [train_char[:1500], train_char[1500:1900] + ["", "zz"], [], train_char[1900:2000]]
    nat = natw = py = pyw = None
    py, pyw = {}, {}
    for lines in rounds:
        nat, natw = secondary_counter(list(lines), "", "\x03", "\x00", 0, 1, 256, 3, nwords_dict=nat, words=natw)
        sec_mod.IMPORT_PLAIN_COUNTS = False            # the dict-based trainer, as the comparison
        try:
            py, pyw = secondary_counter(list(lines), "", "\x03", "\x00", 0, 1, 256, 3, nwords_dict=py, words=pyw)
        finally:
            sec_mod.IMPORT_PLAIN_COUNTS = True
        assert isinstance(nat, NativeCounts) and not isinstance(py, NativeCounts)
        assert [(k, list(v.items())) for k, v in nat.items()] == [(k, list(v.items())) for k, v in py.items()]
        assert list(natw.items()) == list(pyw.items())
        cfg = {"start_chr": "\x03", "end_chr": "\x00", "threshold": 3}
        m = BackWordsSecondaryMonteCarlo((nat, natw, cfg))
        assert hasattr(m, "_trained")
        assert rows_of(m.nwords) == rows_of(expand_2d(freq2prob(py, 3)))
    back = pickle.loads(pickle.dumps((nat, natw, cfg)))
    assert type(back[0]) is dict and back[0] == py


def test_reference_pickle_resumes_natively(train_char):
    """A model pickle written by the REFERENCE's secondary trainer (tests/golden/make_golden_pickle.py:
    plain dicts) loads into native counts (mc_counts_import): the simulator built from it has the
    reference's round-0 table, the next round merges in C++ and ends at the reference's round-1 counts
    and table; a pickle of ours resumes the same way; dicts that are not a trainer's stay dicts."""
    import pickle
    from conftest import GOLDEN
    from montecarlo_amd.trained import NativeCounts
    fx = golden("secondary_thr4_char.json")
    with open(os.path.join(GOLDEN, "secondary_round0_reference.pickle"), "rb") as fd:
        m = BackWordsSecondaryMonteCarlo(fd)
    assert hasattr(m, "_trained")                       # freq2prob + compile ran in C++
    check_table(fx["round0"], m)
    with open(os.path.join(GOLDEN, "secondary_round0_reference.pickle"), "rb") as fd:
        counts, words, cfg = pickle.load(fd)
    assert type(counts) is dict
    nat = NativeCounts.from_plain(counts, words, cfg["start_chr"], cfg["end_chr"])
    assert [(k, list(v.items())) for k, v in nat.items()] == [(k, list(v.items())) for k, v in counts.items()]
    assert list(nat.words().items()) == list(words.items())
    counts1, words1 = secondary_counter(train_char[2500:3200], "", "\x03", "\x00", 0, 1, 256, 4, nwords_dict=counts,
                                        words=words)
    assert isinstance(counts1, NativeCounts)
    assert digest([[list(k), sorted(v.items())] for k, v in counts1.items()]) == fx["meta"]["counts_digest"]
    # the reference updates the caller's dicts in place (backwords_secondary_trainer.py:44-78): so does this
    assert words1 is words and list(words.items()) == list(counts1.words().items())
    assert [(k, list(v.items())) for k, v in counts.items()] == [(k, list(v.items())) for k, v in counts1.items()]
    m1 = BackWordsSecondaryMonteCarlo((counts1, words1, cfg))
    check_table(fx["round1"], m1)
    # our own pickle of round 0 -> plain dicts -> native again -> same round 1
    back = pickle.loads(pickle.dumps((nat, nat.words(), cfg)))
    counts2, words2 = secondary_counter(train_char[2500:3200], "", "\x03", "\x00", 0, 1, 256, 4, nwords_dict=back[0],
                                        words=back[1])
    assert isinstance(counts2, NativeCounts) and dict(counts2) == dict(counts1) and words2 == words1
    # not a trainer's dicts: the empty-context row disagrees with `words` / a count is not an integer
    bad = {k: dict(v) for k, v in counts.items()}
    bad[()]["a"] = bad[()].get("a", 0) + 1
    assert NativeCounts.from_plain(bad, words, "\x03", "\x00") is None
    odd = {k: dict(v) for k, v in counts.items()}
    odd[next(k for k in odd if k)]["zz"] = "many"
    assert NativeCounts.from_plain(odd, words, "\x03", "\x00") is None
    # ... and the raw entry point refuses inconsistent arrays
    h = ctypes.c_void_p()
    nv.check(nv.lib.mc_counts_new(ctypes.byref(h)))
    off = np.array([0, 1, 2], dtype=np.uint32)
    txt = np.array([3, 0], dtype=np.uint32)
    wsym, wcnt = np.array([0, 7], dtype=np.uint32), np.array([1, 1], dtype=np.uint64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    rc = nv.lib.mc_counts_import(h, vp(off), vp(txt), 2, vp(wsym), vp(wcnt), 2, None, None, None, None, None, 0, 3, 0)
    assert rc < 0 and b"inconsistent" in nv.lib.mc_last_error()
    nv.lib.mc_counts_free(h)


def test_compile_rejects_bad_tables():
    from montecarlo_amd.lib4mc.ProbLib import expand_2d
    from montecarlo_amd.tables import FlatTable
    # ("a","b") is a row but ("a",) is not: the backoff automaton would not be exact
    table = expand_2d({(): {"a": .5, "\x03": .5}, ("a", "b"): {"\x03": 1.0}})
    with pytest.raises(nv.NativeError, match="prefix-closed"):
        FlatTable(table, nv.KIND_BACKOFF, start_units=("\x00",))


def test_native_line_ingest_equals_dict_counting():
    """mc_lines_ingest == the counting loop of parse_file (reference lib4mc/MonteCarloParent.py:56-60):
    unique lines in first-seen order, multiplicities, code points."""
    from montecarlo_amd.ingest import UniqueLines
    rng = np.random.default_rng(7)
    pool = ["", "a", "password", "pässwörd", "密码123", "x" * 300, "\U0001F600ok", "tab\tinside", " lead", "trail ",
            "abcdefgh", "abcdefgh1", "abcdefg", "\ud800lone"] + synth.zipf_corpus(3000, vocab=500)
    lines = [pool[i] for i in rng.integers(0, len(pool), size=20000)]
    for text in ("\n".join(lines) + "\n", "\n".join(lines), "", "\n", "\n\n", "one", "\rcr-lead\r\nx\r\n\r\n"):
        want = {}
        for ln in text.split("\n")[:-1] + ([text.split("\n")[-1]] if not text.endswith("\n") and text else []):
            ln = ln.strip("\r\n")
            want[ln] = want.get(ln, 0) + 1
        got = UniqueLines(text)
        assert got.strings() == list(want)
        assert got.count.tolist() == list(want.values())
        assert got.n_lines == sum(want.values())
        assert got.longest == max((len(k) for k in want), default=0)
        assert got.off.tolist() == np.concatenate([[0], np.cumsum([len(k) for k in want])]).astype(int).tolist()
    with pytest.raises(nv.NativeError):
        h = ctypes.c_void_p()
        nv.check(nv.lib.mc_lines_ingest(b"ok\n\xff\xfe\n", 6, ctypes.byref(h)))


ENUM_CASES = ["nwords_n3_char", "nwords_n4_char_min6", "nwords_n2_tab", "backwords_thr10_char",
              "backwords_thr3_char_min7", "backwords_thr3_tab"]


def enum_model(fx, train_char, train_tab):
    """(probability table, FlatTable as the product's enumerator builds it) for a golden enumerator case."""
    from montecarlo_amd.backwords.backwords_trainer import backwords_counter
    from montecarlo_amd.lib4mc.ProbLib import expand_2d
    from montecarlo_amd.nwords.nwords_trainer import nwords_counter
    from montecarlo_amd.tables import FlatTable
    lines = train_char if fx["train"] == "train_char.txt" else train_tab
    if fx["kind"] == "nwords":
        probs, _ = nwords_counter(synth.as_file(lines), fx["n"], fx["splitter"], fx["end_chr"], 0, 1,
                                  start_chr=fx["start_chr"])
        flat = FlatTable(expand_2d(probs), nv.KIND_NGRAM, order=fx["n"] - 1,
                         start_units=(fx["start_chr"],) * (fx["n"] - 1), end_chr=fx["end_chr"], min_len=fx["min_len"])
    else:
        probs, _ = backwords_counter(synth.as_file(lines), fx["splitter"], fx["start_chr"], fx["end_chr"], 0, 1,
                                     fx["count_threshold"], 256)
        flat = FlatTable(expand_2d(probs), nv.KIND_BACKOFF, order=0, start_units=(fx["start_chr"],),
                         end_chr=fx["end_chr"], min_len=fx["min_len"])
    return probs, flat


def check_enum_text(fx, text):
    lines = text.split("\n")[:-1]
    assert len(lines) == fx["n_lines"]
    if "lines" in fx:
        assert lines == fx["lines"]
    else:
        assert lines[:200] == fx["head"] and lines[-200:] == fx["tail"]
    assert hashlib.sha256(text.encode("utf-8", "surrogatepass")).hexdigest() == fx["sha256"]


@pytest.mark.parametrize("case", ENUM_CASES)
def test_enumeration_walk_over_compiled_table(case, train_char, train_tab):
    """The expansion rule of mc_enumerate (tests/walker.py steps state/payload as k_enum_expand does),
    depth first, writes what the reference's enumerators wrote: pins the next-state links the kernel
    relies on, the end symbol's included."""
    fx = golden("enumerators.json")[case]
    _, flat = enum_model(fx, train_char, train_tab)
    backoff = fx["kind"] == "backwords"
    found = CompiledWalker(flat).enumerate(fx["threshold"], fx["min_len"], 256 if backoff else 30, backoff)
    digits = 5 if backoff else 8
    text = "".join("%s\t%.*f\n" % ("".join(flat.symbols[s] for s in syms), digits, x) for syms, x in found)
    check_enum_text(fx, text)


def test_native_write2_formatting_equals_fstrings():
    """mc_format_gc == the f-string loop of write2 (reference lib4mc/MonteCarloLib.py:65-68) byte for byte:
    rounding ties of %.8f / %5.2f, wide integers, non-ASCII passwords; rows it does not take (components
    lists, guess numbers beyond int64, a newline inside a password, NaN) leave it to the loop."""
    from montecarlo_amd.lib4mc.MonteCarloLib import _format_native
    rng = np.random.default_rng(11)
    n = 5000
    pw = synth.zipf_corpus(n - 6, vocab=2000) + ["", "pässwörd", "密码123", "tab\tinside", "\U0001F600", " x "]
    mlp = (rng.random(n) * 80).tolist()
    mlp[:12] = [0.0, -0.0, 0.5, 1.5, 2.675, 1e-9, 5e-9, 0.123456785, 0.123456775, 1022.0, 1e15, 33.000000005]
    apps = rng.integers(1, 10 ** 6, size=n).tolist()
    rank = [int(x) for x in np.cumsum(rng.integers(1, 10 ** 12, size=n))]
    rank[-1] = 2 ** 63 - 1
    crk = np.cumsum(apps).tolist()
    ratio = [c / crk[-1] * 100 for c in crk]
    ratio[:4] = [0.005, 0.015, 99.995, 100.0]
    gc = list(zip(pw, mlp, apps, rank, crk, ratio))
    want = "".join(f"{p}\t{m:.8f}\t{a}\t{r}\t{c}\t{q:5.2f}\n" for p, m, a, r, c, q in gc)
    assert _format_native(gc) == want
    assert _format_native(gc[:100]) is None                                     # small: not worth the call
    for bad in ((["a", "b"], 1.0, 1, 1, 1, 1.0), ("x", 1.0, 1, 2 ** 63, 1, 1.0), ("x\ny", 1.0, 1, 1, 1, 1.0),
                ("x", float("nan"), 1, 1, 1, 1.0), ("\ud800", 1.0, 1, 1, 1, 1.0)):
        assert _format_native(gc[:600] + [bad]) is None


def test_format_gc_cps_equals_fstrings():
    """mc_format_gc_cps (passwords as the ingest's code points + an order) writes what the reference's f-string
    writes (lib4mc/MonteCarloLib.py:65-66), multi-byte characters included; a lone surrogate is refused."""
    from montecarlo_amd.ingest import UniqueLines
    rng = np.random.default_rng(11)
    alphabet = "abcXYZ019!\u00e9\u4e2d\U0001f600\t "
    pw = ["".join(alphabet[i] for i in rng.integers(0, len(alphabet), rng.integers(0, 12))) + f"#{k}"
          for k in range(700)]
    lines = UniqueLines("\n".join(pw) + "\n")
    assert lines.strings() == pw
    order = rng.permutation(len(pw)).astype(np.uint32)
    n = len(pw)
    mlp = np.sort(rng.random(n) * 200)
    apps = rng.integers(1, 50, n).astype(np.int64)
    rank = np.cumsum(rng.integers(1, 2 ** 40, n)).astype(np.int64)
    crk = np.cumsum(apps)
    ratio = crk / float(crk[-1]) * 100
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    out = np.empty(4 * len(lines.cps) + 120 * n, dtype=np.uint8)
    size = nv.lib.mc_format_gc_cps(vp(lines.cps), vp(lines.off), len(lines.off) - 1, vp(order), n, vp(mlp), vp(apps), vp(rank), vp(crk),
                                   vp(ratio), vp(out), out.size)
    want = "".join(f"{pw[o]}\t{m:.8f}\t{a}\t{r}\t{c}\t{q:5.2f}\n"
                   for o, m, a, r, c, q in zip(order.tolist(), mlp.tolist(), apps.tolist(), rank.tolist(),
                                               crk.tolist(), ratio.tolist()))
    assert size == len(want.encode("utf-8")) and bytes(out[:size]).decode("utf-8") == want
    assert nv.lib.mc_format_gc_cps(vp(lines.cps), vp(lines.off), len(lines.off) - 1, vp(order), n, vp(mlp), vp(apps), vp(rank), vp(crk),
                                   vp(ratio), vp(out), 100) < 0                     # dst too small
    bad = UniqueLines("ab\n\ud800x\n")
    two = np.arange(2, dtype=np.uint32)
    assert nv.lib.mc_format_gc_cps(vp(bad.cps), vp(bad.off), len(bad.off) - 1, vp(two), 2, vp(mlp), vp(apps), vp(rank), vp(crk),
                                   vp(ratio), vp(out), out.size) < 0


def test_column_lists_behave_as_the_lists_they_replace():
    """ScoredList / GcList (montecarlo_amd.columns) against plain lists of tuples: length, indexing, slicing,
    iteration, equality, and that a change through the list interface turns `pristine` off."""
    from montecarlo_amd.columns import GcList, ScoredList
    from montecarlo_amd.ingest import UniqueLines
    pw = [f"pw{k}" for k in range(50)]
    lines = UniqueLines("\n".join(pw + pw[:7]) + "\n")
    ml2p = np.linspace(1, 20, 50)
    order = np.arange(50, dtype=np.uint32)[::-1].copy()
    sc = ScoredList(order, lines.count[order].astype(np.int64), ml2p, lines=lines)
    want = [(pw[o], int(lines.count[o]), float(m)) for o, m in zip(order.tolist(), ml2p.tolist())]
    assert len(sc) == 50 and sc.pristine
    assert sc == want and want == list(sc) and sc[3] == want[3] and sc[-2:] == want[-2:]
    assert [type(v) for v in sc[0]] == [str, int, float]
    ranks = np.arange(1, 51, dtype=np.int64) * 3
    crk = np.cumsum(sc.count)
    gcl = GcList(sc, ranks, crk, crk / float(crk[-1]) * 100)
    rows = [(p, m, a, r, c, c / int(crk[-1]) * 100) for (p, a, m), r, c in zip(want, ranks.tolist(), crk.tolist())]
    assert gcl == rows and gcl.pristine and len(gcl) == 50
    # concatenation from either side, pickling, copying and JSON: plain lists of the rows (columns.py, module text)
    import copy
    import json
    import pickle
    head = [("x", 1, 0.5)]
    assert head + sc == head + want and sc + head == want + head and type(head + sc) is list
    assert pickle.loads(pickle.dumps(gcl)) == rows and type(pickle.loads(pickle.dumps(sc))) is list
    assert copy.copy(sc) == want and copy.deepcopy(gcl) == rows
    with pytest.raises(TypeError):
        json.dumps(sc)
    assert json.loads(json.dumps(sc, default=list)) == [list(r) for r in want]
    assert json.loads(json.dumps(list(gcl))) == [list(r) for r in rows]
    gcl[0] = ("zz", 1.0, 1, 1, 1, 1.0)
    assert not gcl.pristine and gcl[0][0] == "zz" and gcl[1:] == rows[1:]
    sc.sort(key=lambda r: r[0])
    assert not sc.pristine and sc == sorted(want, key=lambda r: r[0])


def test_c_abi_rejects_bad_arguments_before_touching_the_device():
    """Argument checks of the launch entry points come before any HIP call, so they can be exercised
    on a host without a GPU: null handles / buffers, aliasing outputs, sizes beyond the documented
    limits, a workspace that is too small.  Every rejection sets mc_last_error()."""
    lib = nv.lib
    buf = np.zeros(64, dtype=np.float64)
    a, b, c = (ctypes.c_void_p(buf.ctypes.data + 8 * i) for i in (0, 16, 32))
    big = ctypes.c_void_p(buf.ctypes.data + 8 * 48)

    def rejected(rc, what):
        assert rc < 0
        with pytest.raises(nv.NativeError, match=what):
            nv.check(rc)

    need = lib.mc_rank_workspace_bytes(10)
    assert need >= 10 * 8 and lib.mc_rank_workspace_bytes(10 ** 7) > lib.mc_rank_workspace_bytes(10 ** 6) > need
    assert lib.mc_rank_build(None, 0, 0, None, None, None, 0, None) == 0            # nothing to do
    rejected(lib.mc_rank_build(None, 10, 10, b, c, big, need, None), "mc_rank_build")
    rejected(lib.mc_rank_build(a, 10, 9, b, c, big, need, None), "mc_rank_build")    # n_global < n
    rejected(lib.mc_rank_build(a, 10, 10, a, c, big, need, None), "mc_rank_build")   # sorted in place
    rejected(lib.mc_rank_build(a, 10, 10, b, b, big, need, None), "mc_rank_build")   # positions over the keys
    rejected(lib.mc_rank_build(a, 2 ** 32, 2 ** 32, b, c, big, 1 << 62, None), "2\\^32")
    rejected(lib.mc_rank_build(a, 10, 10, b, c, big, need - 1, None), "workspace")
    assert lib.mc_rank_query(None, None, 0, None, 0, None, None, None) == 0
    assert lib.mc_order_keys(None, 0, None, None, None, 0, None) == 0
    oneed = lib.mc_order_workspace_bytes(10)
    rejected(lib.mc_order_keys(a, 10, a, c, big, oneed, None), "mc_order_keys")       # sorted over the input
    rejected(lib.mc_order_keys(a, 10, b, None, big, oneed, None), "mc_order_keys")
    rejected(lib.mc_order_keys(a, 2 ** 32, b, c, big, 1 << 62, None), "2\\^32")
    rejected(lib.mc_order_keys(a, 10, b, c, big, oneed - 1, None), "workspace")
    rejected(lib.mc_guess_numbers(a, None, 10, 1, b, None, big, 1 << 20, None), "mc_guess_numbers")
    h = ctypes.c_void_p()
    rejected(lib.mc_comm_init(None, 0, 1, ctypes.byref(h)), "mc_comm_init")
    rejected(lib.mc_comm_init(a, 2, 2, ctypes.byref(h)), "mc_comm_init")             # rank outside the world
    rejected(lib.mc_comm_unique_id(None), "mc_comm_unique_id")
    rejected(lib.mc_allreduce_rank(None, a, None, 10, None), "mc_allreduce_rank")
    lib.mc_comm_destroy(None)                                                        # nothing to do
    rejected(lib.mc_rank_query(a, b, 10, None, 5, c, None, None), "mc_rank_query")
    rejected(lib.mc_rank_query(None, b, 10, a, 5, c, None, None), "mc_rank_query")
    rejected(lib.mc_sample(None, 0, 10, 1, 100, a, None, None, None, 0, None), "mc_sample")
    rejected(lib.mc_sample_injected(None, 10, a, 4, b, None, None, None, 0, None), "mc_sample_injected")
    rejected(lib.mc_score(None, a, b, 3, c, None), "mc_score")
    rejected(lib.mc_score_bpe(None, a, a, b, 3, c, None), "mc_score_bpe")
    rejected(lib.mc_score_words(None, a, b, 3, 30, 10, c, None, 0, big, 64, None), "mc_score_words")
    assert lib.mc_score_words_workspace_bytes(None, 10, 30) == 0
    rejected(lib.mc_model_check(None, None, None), "mc_model_check")
    # the table of the dedup must be a power of two of at least twice the rows
    rejected(lib.mc_sample_dedup(a, a, 8, 10, 16, a, b, c, big, None), "power of two")
    rejected(lib.mc_sample_dedup(a, a, 8, 10, 24, a, b, c, big, None), "power of two")
    rejected(lib.mc_model_upload(None, 0, ctypes.byref(ctypes.c_void_p())), "mc_model_upload")
    lib.mc_model_destroy(None)                                                       # a no-op, as free(NULL)


def test_two_choice_symbol_table_holds_every_symbol_once(train_tab):
    """MC_ARR_WORDCK (the symbol-text map as a two-choice table, read by k_score_words_fast / k_score_bpe_fast in one
    round of loads) against the symbols themselves, by the header's own formulas: every symbol with a text sits in
    one of its two slots with its tag, in no other slot, and the table holds nothing else; texts that are no symbol
    find no slot whose tag AND text match (a reader compares the text after a tag hit).  On a sub-word backoff
    model, a single-character model and a sub-word model of 3000 chunks."""
    models = [BackWordsMonteCarlo(synth.as_file(train_tab), splitter="\t", threshold=3),
              BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(3000, vocab=300, seed=5)), splitter="", threshold=5),
              BackWordsMonteCarlo(synth.as_file(synth.subword_corpus(20000)), splitter="\t", threshold=10)]
    M64 = (1 << 64) - 1
    for m in models:
        flat = m._flat
        ck = flat.array(nv.ARR_WORDCK)
        off, text = flat.array(nv.ARR_SYMTEXTOFF), flat.array(nv.ARR_SYMTEXT)
        cap = len(ck)
        assert cap >= 16 and cap & (cap - 1) == 0
        n_sym = len(off) - 1
        assert cap >= 4 * sum(1 for s in range(n_sym) if off[s + 1] > off[s])

        def slots(cps):
            arr = np.ascontiguousarray(cps, dtype=np.uint32)
            h = int(nv.lib.mc_text_hash(arr.ctypes.data_as(nv.u32p), len(arr)))
            p2 = ((((h ^ (h >> 29)) * 0x9E3779B97F4A7C15) & M64) >> 21) & (cap - 1)
            tag = ((h >> 32) ^ ((len(arr) * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF
            return h & (cap - 1), p2, tag

        held = {}
        for s in range(n_sym):
            cps = text[off[s]:off[s + 1]]
            if len(cps) == 0:
                continue
            p1, p2, tag = slots(cps)
            want = (tag << 32) | s
            at = [p for p in {p1, p2} if int(ck[p]) == want]
            assert len(at) == 1, (s, p1, p2)
            held[at[0]] = want
        occupied = {i: int(v) for i, v in enumerate(ck.tolist()) if int(v) != M64}
        assert occupied == held
        # texts that are no symbols: a tag hit (if any) names a symbol of another text
        rng = np.random.default_rng(2)
        known = {tuple(text[off[s]:off[s + 1]].tolist()) for s in range(n_sym)}
        for _ in range(3000):
            cps = tuple(int(c) for c in rng.integers(33, 126, size=int(rng.integers(1, 7))))
            if cps in known:
                continue
            p1, p2, tag = slots(cps)
            for p in (p1, p2):
                v = int(ck[p])
                if v != M64 and (v >> 32) == tag:
                    s = v & 0xFFFFFFFF
                    assert tuple(text[off[s]:off[s + 1]].tolist()) != cps
