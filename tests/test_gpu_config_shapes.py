"""Parity at the SHAPE of BASELINE configs 3 and 4 and of the large sub-word model (run with -m gpu).

The golden fixtures pin these kernels on toy models (BPE rows of <= 5 terminals, v2 n=4 on 3,000 lines);
here the same kernels run on models of the size the configs name -- BPE-PCFG terminal rows of up to 1e5
entries with mixed-class rows, the 100k-line v2 6-gram model, the 100k-line tab-split backoff model with
6.3 M entries -- and are compared bit for bit with the CPU restatements (oracle/mc_oracle.py,
oracle/mc_oracle.c), which tests/test_oracle_golden.py pins against the reference's own outputs.
Independent chains on both sides: the oracle trains / reads the model with its own code, the product with
the native trainers and table compiler.  Reference: bpe_simulator.py:105-150, nwords_simulator_v2.py:60-82,
nwords_simulator.py:36-60, backwords_simulator.py:28-49."""
import tempfile

import numpy as np
import torch
import pytest

from conftest import golden, hx  # noqa: F401
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.nwords_simulator import NWordsMonteCarlo
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
from oracle import c_oracle
from oracle import mc_oracle as orc
from oracle.philox import SampleStream, uniform53

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_bpe_config4_shape():
    """k_sample_bpe / k_score_bpe on a model of config 4's shape: 400 structures, terminal rows of 10..1e5
    entries (the draw table's multi-threshold buckets and the word / label hashes at realistic load), mixed
    DM/TM/FM rows (wildcard labels, not_parsed fallback).  1e5 samples: ml2p bits, strings, draw counts;
    scores of the sampled strings, of mutated ones, of out-of-model ones and of the empty string."""
    from montecarlo_amd.bpe_simulator import BpePcfgSim
    with tempfile.TemporaryDirectory() as tmp:
        synth.zipf_bpe_dir(tmp, np.random.default_rng(4), mixed=True)
        m = BpePcfgSim(tmp)
        o = orc.BpeOracle(*orc.read_bpe(tmp))
    assert max(len(row[1]) for row in m.terminals.values()) == 100000
    n, seed, first = 100_000, 0xB9E, 7_000_000_000
    res = m._dev.sample(n, seed, first_id=first, want_draws=True, want_len=True, sym_stride=m._flat.max_segments)
    cnt = m._dev.check()
    got_s = m._flat.decode_entries(res["syms"].cpu().numpy(), res["len"].cpu().numpy())
    want_x, want_s, want_d = [], [], []
    for i in range(n):
        st = SampleStream(seed, first + i)
        x, s = o.sample1(st)
        want_x.append(x)
        want_s.append(s)
        want_d.append(st.draws)
    assert np.array_equal(bits(res["ml2p"].cpu().numpy()), bits(want_x))
    assert got_s == want_s
    assert np.array_equal(res["draws"].cpu().numpy().astype(np.int64), np.array(want_d))
    assert cnt["draws"] == sum(want_d) and cnt["failed"] == 0
    assert cnt["indirect_draws"] > 0                         # buckets with thresholds inside were exercised
    # injected-uniform mode on the same model, 2000 samples
    k = 2000
    dmax = max(want_d[:k])
    u = np.array([[uniform53(seed, first + i, d) for d in range(dmax)] for i in range(k)])
    inj = m._dev.sample_injected(u, sym_stride=m._flat.max_segments)
    m._dev.check()
    assert np.array_equal(bits(inj["ml2p"].cpu().numpy()), bits(want_x[:k]))
    assert m._flat.decode_entries(inj["syms"].cpu().numpy(), inj["len"].cpu().numpy()) == want_s[:k]
    # scoring: sampled strings, strings with one character changed (most leave their terminal row or their
    # label: other candidates, not_parsed, or no parse), strings of classes the model never joins, empty
    rng = np.random.default_rng(9)
    uniq = list(dict.fromkeys(want_s))
    alphabet = "abcxyzABC0189!@._~ é"
    mutated = []
    for s in uniq[:30000]:
        if s:
            j = int(rng.integers(0, len(s)))
            mutated.append(s[:j] + alphabet[int(rng.integers(0, len(alphabet)))] + s[j + 1:])
    pw = uniq + mutated + ["", "~~~~", "a" * 40, "Zz9!Zz9!Zz9!", "é", "1a2b3c4d"]
    got, parts = m.score_batch(pw)
    want = [o.calc_ml2p(p) for p in pw]
    assert np.array_equal(bits(got), bits(want))
    assert parts[0] == [pw[0]]
    unparsed = sum(1 for x in want if x == orc.UNPARSEABLE)
    assert 100 < unparsed < len(pw) - len(uniq) + 1
    # the wildcard labels and the not_parsed fallback were reached with a parse found
    reached = sum(1 for p, x in zip(pw, want)
                  if x < orc.UNPARSEABLE and orc.luds(p) not in o.converted)
    assert reached > 0
    # both scoring kernels on the same batch (k_score_bpe_fast is what ran above; mc_score_bpe_debug_kernel(1): the
    # walk, one load after another), with passwords at and past the fast kernel's 32 characters
    from montecarlo_amd import _native as nv
    longer = [(s * 6)[:n_] for s in uniq[:300] if s for n_ in (31, 32, 33, 47)]
    cps, cls, off = m.encode(pw + longer)
    nv.lib.mc_score_bpe_debug_kernel(1)
    try:
        walk = m._dev.score_bpe(cps, cls, off).cpu().numpy()
    finally:
        nv.lib.mc_score_bpe_debug_kernel(0)
    fast = m._dev.score_bpe(cps, cls, off).cpu().numpy()
    assert np.array_equal(bits(fast), bits(walk))
    assert np.array_equal(bits(fast[:len(pw)]), bits(want))
    assert np.array_equal(bits(fast[len(pw):]), bits([o.calc_ml2p(p) for p in longer]))


def test_bpe_certain_structure_samples_plus_zero():
    """A structure of probability 1 over terminals of probability 1: every stored -log2 p is -0.0, and the reference's
    sum starts at +0.0 (`prob = .0; prob += p`, bpe_simulator.py:107-109), so a sample's key is +0.0 -- not -0.0, whose
    bits are the largest of all in the key sort.  (scripts/model_fuzz.py found k_sample_bpe starting at the first value.)"""
    import os
    from montecarlo_amd.bpe_simulator import BpePcfgSim
    from montecarlo_amd.engine import DeviceRank
    with tempfile.TemporaryDirectory() as tmp:
        for sub in ("grammar", "lower", "upper", "mixed_2", "mixed_3", "mixed_4", "digits", "special"):
            os.makedirs(os.path.join(tmp, sub))
        with open(os.path.join(tmp, "grammar", "structures.txt"), "w") as f:
            f.write("L2D1\t1.0\n")
        with open(os.path.join(tmp, "lower", "2.txt"), "w") as f:
            f.write("ab\t1.0\n")
        with open(os.path.join(tmp, "digits", "1.txt"), "w") as f:
            f.write("7\t1.0\n")
        m = BpePcfgSim(tmp)
        o = orc.BpeOracle(*orc.read_bpe(tmp))
    res = m._dev.sample(5000, 3, first_id=11, want_len=True, sym_stride=m._flat.max_segments)
    assert m._dev.check()["failed"] == 0
    keys = res["ml2p"]
    assert bool((keys.view(torch.int64) == 0).all())                       # +0.0, bit for bit
    x, s = o.sample1(SampleStream(3, 11))
    assert bits([x])[0] == 0 and s == "ab7"
    assert set(m._flat.decode_entries(res["syms"].cpu().numpy(), res["len"].cpu().numpy())) == {"ab7"}
    rk = DeviceRank(torch.cat([keys, torch.full((5,), 2.5, dtype=torch.float64, device=keys.device)]))
    assert rk.check()["error"] == 0 and float(rk.keys_sorted[0]) == 0.0 and float(rk.keys_sorted[-1]) == 2.5
    got, _ = m.score_batch(["ab7", "ab8"])
    assert bits(got)[0] == bits([o.calc_ml2p("ab7")])[0] and got[1] == orc.UNPARSEABLE


def test_v2_n6_config3_model_matches_c_oracle():
    """k_sample on the v2 6-gram model of config 3 (100k synthetic lines, character-suffix contexts, word
    transitions, the pwd_len-survives-restart quirk): 2e5 samples at ids beyond 2^32 against
    oracle/mc_oracle.c (KIND_V2) trained by the oracle's own train_v2."""
    lines = synth.v2_corpus(100000)
    m = NWords2MonteCarlo(synth.as_file(lines), 6)
    probs, _words = orc.train_v2(lines, 6)
    cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_V2, order=5, start_units=[], flags=1)
    n, seed, first = 200_000, 77, 5_000_000_000
    res = m._dev.sample(n, seed, first_id=first, want_len=True, want_draws=True, sym_stride=24)
    cnt = m._dev.check()
    want = cm.sample(n, seed, first_id=first, stride=24)
    assert np.array_equal(bits(res["ml2p"].cpu().numpy()), bits(want["ml2p"]))
    assert np.array_equal(res["draws"].cpu().numpy().astype(np.uint32), want["draws"])
    assert np.array_equal(res["len"].cpu().numpy().astype(np.uint32), want["len"])
    assert cnt["draws"] == int(want["draws"].sum()) and cnt["failed"] == 0 and cnt["restarts"] > 0
    got_s = m._flat.decode(res["syms"].cpu().numpy(), np.minimum(res["len"].cpu().numpy(), 24))
    want_s = cm.strings({"syms": want["syms"], "len": np.minimum(want["len"], 24)})
    assert got_s == want_s
    # injected uniforms on the multi-character alphabet (k_sample<injected, multichar>)
    k = 3000
    dmax = int(want["draws"][:k].max())
    u = np.array([[uniform53(seed, first + i, d) for d in range(dmax)] for i in range(k)])
    inj = m._dev.sample_injected(u, sym_stride=24)
    m._dev.check()
    assert np.array_equal(bits(inj["ml2p"].cpu().numpy()), bits(want["ml2p"][:k]))
    assert np.array_equal(inj["draws"].cpu().numpy().astype(np.uint32), want["draws"][:k])


def test_subword_dfs_scoring_large_model():
    """k_score_words on the 100k-line tab-split backoff model (2,441 rows, 6.3 M entries): >= 2e4 distinct
    passwords, most with several competing segmentations, against BackoffOracle.calc_ml2p (values bit for
    bit AND component lists), with the default iter_max and with --max-iter 1 and 3."""
    sub = synth.subword_corpus(100000)
    probs, words = orc.train_backwords(sub, "\t", "\x00", "\x03", 0, 1, 10, 256)
    test = list(dict.fromkeys("".join(line.split("\t")) for line in synth.subword_corpus(60000, stream=1)))
    rng = np.random.default_rng(3)
    extra = []
    for s in test[:3000]:                                     # out-of-model variants: no parse, or another one
        j = int(rng.integers(0, len(s)))
        extra.append(s[:j] + "~" + s[j:])
        extra.append(s + s[: 1 + j % 3])
    pw = test[:22000] + extra + ["", "a", "\x03"]
    changed = 0
    for max_iter in (10 ** 100, 1, 3):
        m = BackWordsMonteCarlo(synth.as_file(sub), splitter="\t", threshold=10, max_iter=max_iter)
        o = orc.BackoffOracle(probs, words, max_iter=max_iter)
        got, parts = m.score_batch(pw)
        want = [o.calc_ml2p(p) for p in pw]
        assert np.array_equal(bits(got), bits([w[0] for w in want])), max_iter
        assert parts == [w[1] for w in want], max_iter
        if max_iter == 10 ** 100:
            full = [w[0] for w in want]
            assert 0 < sum(1 for x in full if x == orc.UNPARSEABLE) < len(extra)
        else:
            changed += sum(1 for a, b in zip(full, want) if a != b[0])
    assert changed > 100          # many passwords have more than one full parse: iter_max decides the result


def test_injected_uniforms_on_multichar_alphabets(train_tab):
    """mc_sample_injected on sub-word models (k_sample<injected, multichar>): same strings / ml2p / draw counts
    as the reference's sampler under the same uniforms (golden)."""
    from test_gpu_parity import device_samples
    m = NWordsMonteCarlo(synth.as_file(train_tab), n=3, splitter="\t")
    device_samples(m, golden("nwords_n3_tab.json"), n=300, injected=True)
    m = BackWordsMonteCarlo(synth.as_file(train_tab), splitter="\t", threshold=3)
    device_samples(m, golden("backwords_thr3_tab.json"), n=300, injected=True)


def test_subword_scoring_kernels_agree():
    """mc_score_words' two kernels on the same inputs: k_score_words_fast (short passwords: characters in LDS, one
    round of loads per position from the two-choice symbol table, tail calls) against k_score_words (the DFS as
    written, mc_score_words_debug_kernel(1)): values bit for bit, split masks, the counters of transition probes and
    steps -- for iter_max 1, 2, 7 and unbounded, with passwords up to the fast kernel's limit (31 characters + the
    end symbol), past it (the launch falls back for the whole batch), empty, and out of the model."""
    from montecarlo_amd import _native as nv
    sub = synth.subword_corpus(30000)
    m = BackWordsMonteCarlo(synth.as_file(sub), splitter="\t", threshold=10)
    base = list(dict.fromkeys("".join(line.split("\t")) for line in synth.subword_corpus(30000, stream=2)))
    rng = np.random.default_rng(11)
    pw = base[:12000] + ["", "a", "~", "\x03"]
    for s in base[:4000]:                                     # longer ones: two or three passwords glued together
        t = s + base[int(rng.integers(0, len(base)))] + base[int(rng.integers(0, len(base)))]
        pw.append(t[:31])
    pw.append((base[0] * 8)[:31])
    for limit, expect_fast in ((31, True), (40, False)):
        batch = pw if limit == 31 else pw + [(base[1] * 10)[:40]]
        cps, off, max_len = m._flat.encode_text(batch)
        assert max_len == limit
        for iter_max in (1, 2, 7, 10 ** 35):
            res = {}
            for which in (0, 1):
                nv.lib.mc_score_words_debug_kernel(which)
                try:
                    m._dev.check()
                    ml2p, split = m._dev.score_words(cps, off, max_len, iter_max)
                    cnt = m._dev.check()
                finally:
                    nv.lib.mc_score_words_debug_kernel(0)
                res[which] = (ml2p.cpu().numpy(), split.cpu().numpy(), cnt["score_probes"], cnt["score_steps"])
            assert np.array_equal(bits(res[0][0]), bits(res[1][0])), (limit, iter_max)
            assert np.array_equal(res[0][1], res[1][1]), (limit, iter_max)
            assert res[0][2:] == res[1][2:], (limit, iter_max)
            assert (res[0][0] < 1022.0).sum() > 10000
    assert expect_fast is False
