"""Pin the CPU oracle (oracle/mc_oracle.py) against the reference's own outputs
recorded in tests/golden/*.json by tests/golden/make_golden.py.  Bit-exact: fp64
values are compared as hex strings."""
import hashlib
import json

import pytest

from conftest import fh, golden, golden_lines, hx
from oracle import mc_oracle as orc
from oracle.philox import SampleStream


def rows_of(table):
    return [[ctx if isinstance(ctx, str) else list(ctx), list(keys), [hx(vals[k]) for k in keys],
             [hx(c) for c in cum]] for ctx, (vals, keys, cum) in table.items()]


def digest(obj):
    return hashlib.sha256(json.dumps(obj, ensure_ascii=True, separators=(",", ":")).encode()).hexdigest()


def check_model(fx, model, words, check_scores=True):
    rows = rows_of(model.table)
    assert len(rows) == fx["n_ctx"]
    assert sum(len(r[1]) for r in rows) == fx["n_entries"]
    assert rows[:len(fx["rows_head"])] == fx["rows_head"]
    assert digest(rows) == fx["table_digest"]
    if "words_digest" in fx:
        assert digest(sorted(words.items())) == fx["words_digest"]
    if check_scores and "scores" in fx:
        for pwd, want, parts in fx["scores"]:
            got, got_parts = model.calc_ml2p(pwd)
            assert (hx(got), got_parts) == (want, parts), repr(pwd)
    seed = fx["samples"]["seed"]
    for i, (want, pwd, draws) in enumerate(fx["samples"]["items"]):
        st = SampleStream(seed, i)
        got, got_pwd = model.sample1(st)
        assert (hx(got), got_pwd, st.draws) == (want, pwd, draws), i


@pytest.mark.parametrize("n", [2, 4])
def test_nwords_char(train_char, n):
    fx = golden(f"nwords_n{n}_char.json")
    probs, words = orc.train_nwords(train_char, n, "", "\x03")
    check_model(fx, orc.NGramOracle(probs, words, n), words)


def test_nwords_tab(train_tab):
    fx = golden("nwords_n3_tab.json")
    probs, words = orc.train_nwords(train_tab, 3, "\t", "\x03")
    check_model(fx, orc.NGramOracle(probs, words, 3), words)


@pytest.mark.parametrize("thr", [2, 10])
def test_backwords_char(train_char, thr):
    fx = golden(f"backwords_thr{thr}_char.json")
    probs, words = orc.train_backwords(train_char, "", "\x00", "\x03", 0, 1, thr, 256)
    check_model(fx, orc.BackoffOracle(probs, words), words)


def test_backwords_tab(train_tab):
    fx = golden("backwords_thr3_tab.json")
    probs, words = orc.train_backwords(train_tab, "\t", "\x00", "\x03", 0, 1, 3, 256)
    check_model(fx, orc.BackoffOracle(probs, words), words)
    fx2 = golden("backwords_thr3_tab_maxiter2.json")
    m = orc.BackoffOracle(probs, words, max_iter=2)
    for pwd, want, parts in fx2["scores"]:
        got, got_parts = m.calc_ml2p(pwd)
        assert (hx(got), got_parts) == (want, parts), repr(pwd)


def test_secondary(train_char):
    fx = golden("secondary_thr4_char.json")
    counts, words = orc.merge_secondary_counts(train_char[:2500], "", "\x03", "\x00", 0, 1, 256, 4)
    m = orc.BackoffOracle(orc.freq2prob(counts, 4), words, start_chr="\x03", end_chr="\x00")
    check_model(fx["round0"], m, words)
    counts, words = orc.merge_secondary_counts(train_char[2500:3200], "", "\x03", "\x00", 0, 1, 256, 4, counts, words)
    assert digest([[list(k), sorted(v.items())] for k, v in counts.items()]) == fx["meta"]["counts_digest"]
    m = orc.BackoffOracle(orc.freq2prob(counts, 4), words, start_chr="\x03", end_chr="\x00")
    check_model(fx["round1"], m, words)


def test_v2(train_v2):
    fx = golden("v2_n4.json")
    probs, words = orc.train_v2(train_v2, 4)
    check_model(fx, orc.V2Oracle(probs, words, 4), words)


def test_bpe():
    import os
    from conftest import GOLDEN
    fx = golden("bpe.json")
    m = orc.BpeOracle(*orc.read_bpe(os.path.join(GOLDEN, "bpe_model")))
    for pwd, want in fx["scores"]:
        assert hx(m.calc_ml2p(pwd)) == want, pwd
    for i, (want, pwd, draws) in enumerate(fx["samples"]["items"]):
        st = SampleStream(fx["samples"]["seed"], i)
        got, got_pwd = m.sample1(st)
        assert (hx(got), got_pwd, st.draws) == (want, pwd, draws), i


def test_rank(train_char):
    fx = golden("rank.json")
    est = orc.RankOracle([fh(x) for x in fx["draw_order"]])
    assert [hx(x) for x in est.keys] == fx["sorted"]
    assert [hx(x) for x in est.positions] == fx["positions"]
    for x, want in fx["probes"]:
        assert hx(est.ml2p2rank(fh(x))) == want
    scored = [(p, c, fh(x)) for p, c, x in fx["scored"]]
    gc = est.iter2gc(scored)
    assert [[p, hx(x), c, str(r), k, hx(ratio)] for p, x, c, r, k, ratio in gc] == fx["gc"]
    assert orc.RankOracle.format_gc(gc) == fx["write2"]
    assert [str(t[3]) for t in est.iter2gc(scored, add1=False)] == fx["gc_add0_ranks"]
    # parse_file restatement: dedup + score + stable sort reproduces the reference's scored list
    probs, words = orc.train_backwords(train_char, "", "\x00", "\x03", 0, 1, 10, 256)
    m = orc.BackoffOracle(probs, words)
    mine = orc.parse_lines(m, golden_lines("test_char.txt"))
    assert [[p, c, hx(x)] for p, c, x in mine] == fx["scored"]


def test_pick_clamps_at_total():
    # reference raises IndexError when u == cum[-1] (ProbLib.py:41-42); the build clamps.
    row = orc.expand_row({"a": 0.5, "b": 0.75})          # total 1.25 > 1
    assert orc.pick(row, lambda: 1.0)[1] == "b"


# ------------------------------------------------------------------ C restatement (oracle/mc_oracle.c)
def _c_model(kind, table, **kw):
    from oracle import c_oracle
    return c_oracle.COracleModel(table, kind, **kw)


def _check_c(fx, cm, drop, n_scores=True):
    items = fx["samples"]["items"]
    res = cm.sample(len(items), fx["samples"]["seed"], stride=300)
    got = [[hx(a), s, int(d)] for a, s, d in zip(res["ml2p"], cm.strings(res, drop=drop), res["draws"])]
    assert got == items
    if n_scores and "scores" in fx:
        pw = [s[0] for s in fx["scores"]]
        assert [hx(x) for x in cm.score(pw)] == [s[1] for s in fx["scores"]]


@pytest.mark.parametrize("n", [2, 4])
def test_c_oracle_nwords(train_char, n):
    from oracle import c_oracle
    probs, words = orc.train_nwords(train_char, n, "", "\x03")
    cm = _c_model(c_oracle.KIND_NGRAM, orc.expand_table(probs), order=n - 1, start_units=["\x00"] * (n - 1))
    _check_c(golden(f"nwords_n{n}_char.json"), cm, "\x00")


@pytest.mark.parametrize("thr", [2, 10])
def test_c_oracle_backwords(train_char, thr):
    from oracle import c_oracle
    probs, words = orc.train_backwords(train_char, "", "\x00", "\x03", 0, 1, thr, 256)
    cm = _c_model(c_oracle.KIND_BACKOFF, orc.expand_table(probs), start_units=["\x00"])
    _check_c(golden(f"backwords_thr{thr}_char.json"), cm, "\x00")


def test_c_oracle_subword_and_v2(train_tab, train_v2):
    from oracle import c_oracle
    probs, _ = orc.train_backwords(train_tab, "\t", "\x00", "\x03", 0, 1, 3, 256)
    cm = _c_model(c_oracle.KIND_BACKOFF, orc.expand_table(probs), start_units=["\x00"])
    _check_c(golden("backwords_thr3_tab.json"), cm, "\x00", n_scores=False)
    probs, _ = orc.train_v2(train_v2, 4)
    cm = _c_model(c_oracle.KIND_V2, orc.expand_table(probs), order=3, start_units=[], flags=1)
    _check_c(golden("v2_n4.json"), cm, "")


def test_c_oracle_rank():
    from oracle import c_oracle
    import numpy as np
    fx = golden("rank.json")
    keys = np.array([fh(x) for x in fx["draw_order"]])
    srt, pos = c_oracle.rank_build(keys)
    assert [hx(x) for x in srt] == fx["sorted"]
    want = np.array([fh(x) for x in fx["positions"]])
    assert np.allclose(pos, want, rtol=1e-12, atol=0)       # libm pow vs numpy power: not bit-pinned
    x = np.array([fh(p[0]) for p in fx["probes"]])
    qp, qc = c_oracle.rank_query(srt, want, x)
    for (xx, w), p, c in zip(fx["probes"], qp, qc):
        assert hx(p if c > 0 else 1) == w


def enum_text(found, digits):
    return "".join(f"{pwd}\t{x:.{digits}f}\n" for pwd, x in found)


def check_enum_text(fx, text):
    lines = text.split("\n")[:-1]
    assert len(lines) == fx["n_lines"]
    if "lines" in fx:
        assert lines == fx["lines"]
    else:
        assert lines[:200] == fx["head"] and lines[-200:] == fx["tail"]
    assert hashlib.sha256(text.encode("utf-8", "surrogatepass")).hexdigest() == fx["sha256"]


@pytest.mark.parametrize("case", ["nwords_n3_char", "nwords_n4_char_min6", "nwords_n2_tab", "backwords_thr10_char",
                                  "backwords_thr3_char_min7", "backwords_thr3_tab"])
def test_enumerators(case, train_char, train_tab):
    """oracle.enumerate_strings == what the reference's enumerators wrote (tests/golden/make_golden_enum.py)."""
    fx = golden("enumerators.json")[case]
    lines = train_char if fx["train"] == "train_char.txt" else train_tab
    if fx["kind"] == "nwords":
        probs, _ = orc.train_nwords(lines, fx["n"], fx["splitter"], fx["end_chr"], start_chr=fx["start_chr"])
        found = orc.enumerate_strings(probs, fx["threshold"], fx["start_chr"], fx["end_chr"], fx["min_len"],
                                      order=fx["n"] - 1)
        check_enum_text(fx, enum_text(found, 8))
    else:
        probs, _ = orc.train_backwords(lines, fx["splitter"], fx["start_chr"], fx["end_chr"], 0, 1,
                                       fx["count_threshold"], 256)
        found = orc.enumerate_strings(probs, fx["threshold"], fx["start_chr"], fx["end_chr"], fx["min_len"])
        check_enum_text(fx, enum_text(found, 5))
        assert fx["early_end_lines"] > 0          # the appended-end-symbol behaviour is in the vectors
