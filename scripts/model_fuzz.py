#!/usr/bin/env python3
"""Differential fuzz of the model side (table compiler, k_sample, k_score) on random small corpora: random alphabets
(2 .. 40 characters, some beyond ASCII), 10 .. 5000 lines with repeats, line lengths 1 .. 14; the backoff model at a
random count threshold or the n-gram model at a random order, trained by the package's native trainer and -- apart --
by the oracle's restatement of the reference's trainer (oracle/mc_oracle.py), sampled by the GPU and by the C oracle
(oracle/mc_oracle.c) from the same Philox stream: -log2 p bit for bit, draw counts, lengths, strings; then the corpus'
lines and mutated ones scored by both.  A fifth of the cases are sub-word models (tab-separated tokens of 1 .. 4
characters: calc_ml2p's DFS over segmentations, values and component lists against the Python oracle, with a random
max_iter), a fifth the v2 model at n = 2 .. 6, a fifth BPE-PCFG model
directories (k_sample_bpe / k_score_bpe against BpeOracle).  Usage: model_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")
from montecarlo_amd import _native as nv  # noqa: E402
from montecarlo_amd import synth  # noqa: E402
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo  # noqa: E402
from montecarlo_amd.nwords_simulator import NWordsMonteCarlo  # noqa: E402
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo  # noqa: E402
from oracle import c_oracle  # noqa: E402
from oracle import mc_oracle as orc  # noqa: E402

POOL = list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJ0123456789!@#$%^&*._-+= ") + ["é", "ß", "ж", "中", "🙂"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def corpus(rng):
    k = int(rng.integers(2, 41))
    alpha = [POOL[i] for i in rng.choice(len(POOL), size=k, replace=False)]
    w = 1.0 / np.arange(1, k + 1) ** float(rng.uniform(0.3, 1.6))
    w /= w.sum()
    vocab = int(10 ** rng.uniform(0.7, 3.3))
    words = []
    for _ in range(vocab):
        ln = 1 + min(13, int(rng.geometric(float(rng.uniform(0.12, 0.5)))))
        words.append("".join(alpha[i] for i in rng.choice(k, size=ln, p=w)))
    words += ["".join(alpha[i] for i in rng.choice(k, size=int(rng.integers(4, 9)), p=w)) for _ in range(3)]  # (some of >= min_len)
    n = int(10 ** rng.uniform(1.0, 3.7))
    if rng.random() < 0.12:                                  # a corpus of one to three distinct lines: rows of a single
        words = words[-int(rng.integers(1, 4)):]             # entry, probabilities of exactly 1 (-log2 p = -0.0)
    zw = 1.0 / np.arange(1, len(words) + 1) ** float(rng.uniform(0.5, 1.3))
    zw /= zw.sum()
    return [words[i] for i in rng.choice(len(words), size=n, p=zw)], alpha


def injected_check(rng, m, cm, exp_t, stride, drop=""):
    """-> (samples that ended within the draws given, all equal?): injected uniforms at the rows' thresholds (see main)"""
    thr_set = []
    for ctx in list(exp_t)[:400]:
        cum_r = exp_t[ctx][2]
        thr_set.append(cum_r[:-1] / cum_r[-1])
    thr_set = np.concatenate(thr_set) if thr_set else np.zeros(0)
    ki, di = 1500, 24
    u_inj = rng.integers(0, 2 ** 53, size=(ki, di)).astype(np.float64) * 2.0 ** -53
    if len(thr_set):
        near = thr_set[rng.integers(0, len(thr_set), size=(ki, di))]
        step_u = rng.integers(-1, 2, size=(ki, di))
        near = np.where(step_u < 0, np.nextafter(near, 0.0), np.where(step_u > 0, np.nextafter(near, 1.0), near))
        u_inj = np.where(rng.random((ki, di)) < 0.45, near, u_inj)
    edge = rng.random((ki, di))
    u_inj = np.where(edge < 0.03, 0.0, np.where(edge > 0.97, 1.0 - 2.0 ** -53, u_inj))
    u_inj = np.clip(u_inj, 0.0, 1.0 - 2.0 ** -53)
    inj = m._dev.sample_injected(u_inj, sym_stride=stride)
    try:
        m._dev.check()
    except nv.NativeError as e:                              # (samples that need more than `di` draws: NaN on both sides)
        assert e.code == nv.ERR_DIVERGED, e
    want_i = cm.sample(ki, 0, uniforms=u_inj, stride=stride)
    gi, lost_i = inj["ml2p"].cpu().numpy(), np.isnan(want_i["ml2p"])
    keep_i = np.flatnonzero(~lost_i)
    same = (np.array_equal(np.isnan(gi), lost_i) and np.array_equal(bits(gi[~lost_i]), bits(want_i["ml2p"][~lost_i]))
            and np.array_equal(inj["draws"].cpu().numpy().astype(np.uint32)[keep_i], want_i["draws"][keep_i])
            and np.array_equal(inj["len"].cpu().numpy().astype(np.uint32)[keep_i], want_i["len"][keep_i])
            and m._flat.decode(inj["syms"].cpu().numpy()[keep_i], np.minimum(inj["len"].cpu().numpy()[keep_i], stride))
            == cm.strings({"syms": want_i["syms"][keep_i], "len": np.minimum(want_i["len"][keep_i], stride)}, drop=drop))
    return len(keep_i), same


def subword_corpus(rng):
    """lines of 1 .. 5 tab-separated tokens of 1 .. 4 characters: many passwords have several segmentations"""
    k = int(rng.integers(2, 12))
    alpha = [POOL[i] for i in rng.choice(len(POOL) - 6, size=k, replace=False)]  # (no space: the trainers strip nothing, but keep it plain)
    vocab = int(10 ** rng.uniform(0.5, 2.3)) + 2
    toks = list(dict.fromkeys("".join(alpha[i] for i in rng.integers(0, k, size=int(rng.integers(1, 5)))) for _ in range(vocab)))
    zw = 1.0 / np.arange(1, len(toks) + 1) ** float(rng.uniform(0.4, 1.2))
    zw /= zw.sum()
    n = int(10 ** rng.uniform(1.0, 3.3))
    return ["\t".join(toks[i] for i in rng.choice(len(toks), size=int(rng.integers(1, 6)), p=zw)) for _ in range(n)], alpha


def subword_case(rng, c):
    """-> (ok, text): sampling against the C oracle, calc_ml2p's DFS (value and components) against the Python oracle"""
    lines, alpha = subword_corpus(rng)
    if int(rng.integers(0, 2)):
        thr = int(rng.choice([1, 2, 5, 10]))
        what = f"sub-word backwords thr={thr}"
        try:
            probs, words = orc.train_backwords(lines, "\t", "\x00", "\x03", 0, 1, thr, 256)
        except ValueError:
            try:
                BackWordsMonteCarlo(synth.as_file(lines), splitter="\t", threshold=thr)
                return False, f"case {c:3d} {what}: the oracle's trainer refused the corpus, the package's did not: BAD"
            except Exception as e:  # noqa: BLE001
                return True, f"case {c:3d} {what}: refused by both trainers ({type(e).__name__})"
        max_iter = int(rng.choice([10 ** 100, 1, 2, 5]))
        m = BackWordsMonteCarlo(synth.as_file(lines), splitter="\t", threshold=thr, max_iter=max_iter)
        o = orc.BackoffOracle(probs, words, max_iter=max_iter)
        cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_BACKOFF, start_units=["\x00"])
        what += f" max_iter={min(max_iter, 99)}"
    else:
        n = int(rng.integers(2, 4))
        what = f"sub-word nwords n={n}"
        m = NWordsMonteCarlo(synth.as_file(lines), n=n, splitter="\t")
        probs, words = orc.train_nwords(lines, n, "\t", "\x03")
        o = orc.NGramOracle(probs, words, n)
        cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_NGRAM, order=n - 1, start_units=["\x00"] * (n - 1))
    count, sd, first = int(rng.integers(1, 20000)), int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
    res = m._dev.sample(count, sd, first_id=first, max_draws=2000, want_len=True, want_draws=True, sym_stride=24)
    try:
        m._dev.check()
        lost_any = False
    except nv.NativeError as e:
        assert e.code == nv.ERR_DIVERGED, e
        lost_any = True
    want = cm.sample(count, sd, first_id=first, max_draws=2000, stride=24)
    got_x = res["ml2p"].cpu().numpy()
    lost = np.isnan(want["ml2p"])
    keep = np.flatnonzero(~lost)
    ok = (np.array_equal(np.isnan(got_x), lost) and np.array_equal(bits(got_x[~lost]), bits(want["ml2p"][~lost]))
          and np.array_equal(res["draws"].cpu().numpy().astype(np.uint32), want["draws"]) and lost_any == bool(lost.any())
          and np.array_equal(res["len"].cpu().numpy().astype(np.uint32)[keep], want["len"][keep]))
    got_s = m._flat.decode(res["syms"].cpu().numpy()[keep], np.minimum(res["len"].cpu().numpy()[keep], 24))
    want_s = cm.strings({"syms": want["syms"][keep], "len": np.minimum(want["len"][keep], 24)}, drop="\x00")
    ok = ok and got_s == want_s
    n_inj, ok_inj = injected_check(rng, m, cm, orc.expand_table(probs), 24, drop="\x00")
    ok = ok and ok_inj
    # (the DFS visits every segmentation, as the reference's does: the number of them grows like 1.9^length on a small
    # alphabet, and the Python oracle walks them one by one -- passwords of at most 14 characters here)
    test = [t for t in dict.fromkeys("".join(line.split("\t")) for line in lines) if len(t) <= 14][:250]
    mut = []
    for t in test[:100]:
        j = int(rng.integers(0, len(t)))
        mut.append(t[:j] + alpha[int(rng.integers(0, len(alpha)))] + t[j + (int(rng.integers(0, 2))):])
    test = test + mut + ["", alpha[0] * 12]
    print(f"case {c:3d} {what}: {len(test)} passwords through the DFS ...", flush=True)
    got, parts = m.score_batch(test)
    ref = [o.calc_ml2p(t) for t in test]
    ok2 = np.array_equal(bits(np.array(got)), bits([r[0] for r in ref])) and parts == [r[1] for r in ref]
    return ok and ok2, (f"case {c:3d} {what:34s} lines={len(lines):5d} alphabet={len(alpha):2d} samples={count:5d} "
                        f"lost={int(lost.sum())} sample={'ok' if ok else 'BAD'} score={'ok' if ok2 else 'BAD'}")


def v2_case(rng, c):
    """the v2 model (contexts of characters, transitions of words) on a small synthetic corpus of its own format"""
    n = int(rng.integers(2, 7))
    lines = synth.v2_corpus(int(10 ** rng.uniform(1.5, 3.5)), vocab=int(10 ** rng.uniform(0.7, 2.8)) + 3,
                            seed=int(rng.integers(0, 2 ** 31)))
    m = NWords2MonteCarlo(synth.as_file(lines), n)
    probs, _words = orc.train_v2(lines, n)
    cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_V2, order=n - 1, start_units=[], flags=1)
    count, sd, first = int(rng.integers(1, 20000)), int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
    res = m._dev.sample(count, sd, first_id=first, max_draws=2000, want_len=True, want_draws=True, sym_stride=24)
    try:
        m._dev.check()
        lost_any = False
    except nv.NativeError as e:
        assert e.code == nv.ERR_DIVERGED, e
        lost_any = True
    want = cm.sample(count, sd, first_id=first, max_draws=2000, stride=24)
    got_x = res["ml2p"].cpu().numpy()
    lost = np.isnan(want["ml2p"])
    keep = np.flatnonzero(~lost)
    ok = (np.array_equal(np.isnan(got_x), lost) and np.array_equal(bits(got_x[~lost]), bits(want["ml2p"][~lost]))
          and np.array_equal(res["draws"].cpu().numpy().astype(np.uint32), want["draws"]) and lost_any == bool(lost.any())
          and np.array_equal(res["len"].cpu().numpy().astype(np.uint32)[keep], want["len"][keep]))
    got_s = m._flat.decode(res["syms"].cpu().numpy()[keep], np.minimum(res["len"].cpu().numpy()[keep], 24))
    want_s = cm.strings({"syms": want["syms"][keep], "len": np.minimum(want["len"][keep], 24)})
    ok = ok and got_s == want_s
    n_inj, ok_inj = injected_check(rng, m, cm, orc.expand_table(probs), 24)
    ok = ok and ok_inj
    return ok, (f"case {c:3d} v2 n={n}                             lines={len(lines):5d} samples={count:5d} "
                f"lost={int(lost.sum())} sample={'ok' if ok else 'BAD'}")


def bpe_dir(path, rng):
    """a small random BPE-PCFG model directory (the format bpeX/modelreader.py reads): terminal rows of 1 .. 500
    entries for a random choice of classes and lengths, mixed-class rows in half the cases, 1 .. 40 structures"""
    for sub in ("grammar", "lower", "upper", "mixed_2", "mixed_3", "mixed_4", "digits", "special"):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    alpha = {"lower": "abcdefghij", "upper": "ABCDEF", "digits": "0123456789", "special": "!@#._"}
    tag = {"lower": "L", "upper": "U", "digits": "D", "special": "S", "mixed_2": "DM", "mixed_3": "TM", "mixed_4": "FM"}
    if int(rng.integers(0, 2)):
        both = alpha["lower"] + alpha["digits"]
        alpha.update({"mixed_2": both, "mixed_3": both + alpha["upper"], "mixed_4": both + alpha["special"]})
    have = []
    for sub, chars in alpha.items():
        chars = np.array(list(chars))
        low = {"mixed_2": 2, "mixed_3": 3, "mixed_4": 4}.get(sub, 1)
        for ln in range(low, low + int(rng.integers(1, 5))):
            if int(rng.integers(0, 4)) == 0 and have:
                continue
            n = min(int(10 ** rng.uniform(0, 2.7)), len(chars) ** ln)
            terms = set()
            while len(terms) < n:
                terms.update("".join(r) for r in rng.choice(chars, size=(n, ln)))
            terms = sorted(terms)[:n]
            rng.shuffle(terms)
            w = 1.0 / np.arange(1, n + 1) ** float(rng.uniform(0.2, 1.5))
            w /= w.sum()
            with open(os.path.join(path, sub, f"{ln}.txt"), "w") as f:
                for t, p in zip(terms, w):
                    f.write(f"{t}\t{float(p)!r}\n")
            have.append((tag[sub], ln))
    structs = set()
    want = min(int(rng.integers(1, 41)) if rng.random() > 0.15 else 1, len(have))
    while len(structs) < want:
        k = int(rng.integers(1, 5))
        segs = [have[int(i)] for i in rng.integers(0, len(have), size=k)]
        if all(a[0] != b[0] for a, b in zip(segs, segs[1:])):
            structs.add("".join(f"{t}{n}" for t, n in segs))
    structs = sorted(structs)
    w = 1.0 / np.arange(1, len(structs) + 1) ** float(rng.uniform(0.3, 1.4))
    w /= w.sum()
    with open(os.path.join(path, "grammar", "structures.txt"), "w") as f:
        for s_, p in zip(structs, w):
            f.write(f"{s_}\t{float(p)!r}\n")
    return len(structs)


def bpe_case(rng, c):
    """k_sample_bpe / k_score_bpe against BpeOracle (bpe_simulator.py:105-150 restated) on a random model directory"""
    import tempfile
    from montecarlo_amd.bpe_simulator import BpePcfgSim
    from oracle.philox import SampleStream
    with tempfile.TemporaryDirectory() as tmp:
        ns = bpe_dir(tmp, rng)
        m = BpePcfgSim(tmp)
        o = orc.BpeOracle(*orc.read_bpe(tmp))
    count, sd, first = int(rng.integers(1, 3000)), int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
    res = m._dev.sample(count, sd, first_id=first, want_draws=True, want_len=True, sym_stride=m._flat.max_segments)
    cnt = m._dev.check()
    got_s = m._flat.decode_entries(res["syms"].cpu().numpy(), res["len"].cpu().numpy())
    want_x, want_s, want_d = [], [], []
    for i in range(count):
        st = SampleStream(sd, first + i)
        x, t = o.sample1(st)
        want_x.append(x)
        want_s.append(t)
        want_d.append(st.draws)
    ok = (np.array_equal(bits(res["ml2p"].cpu().numpy()), bits(want_x)) and got_s == want_s
          and np.array_equal(res["draws"].cpu().numpy().astype(np.int64), np.array(want_d)) and cnt["failed"] == 0)
    if not ok:                                               # (what differs, for whoever reads the log)
        gx, gd = res["ml2p"].cpu().numpy(), res["draws"].cpu().numpy()
        shown = 0
        for i in range(count):
            if (bits([gx[i]])[0] != bits([want_x[i]])[0] or got_s[i] != want_s[i] or int(gd[i]) != want_d[i]) and shown < 4:
                print(f"   sample {i} (id {first + i}): oracle ({want_x[i]!r}, {want_s[i]!r}, {want_d[i]} draws), "
                      f"here ({gx[i]!r}, {got_s[i]!r}, {int(gd[i])} draws)", flush=True)
                shown += 1
        print(f"   counters {cnt}; seed {sd}", flush=True)
    # injected uniforms at the thresholds of the structures' row and of the terminal rows (see the character-level case)
    rows = [o.grammars] + list(o.terminals.values())[:200]
    thr_set = np.concatenate([r[2][:-1] / r[2][-1] for r in rows])
    ki, di = 600, m._flat.max_segments + 1
    u_inj = rng.integers(0, 2 ** 53, size=(ki, di)).astype(np.float64) * 2.0 ** -53
    if len(thr_set):
        near = thr_set[rng.integers(0, len(thr_set), size=(ki, di))]
        step_u = rng.integers(-1, 2, size=(ki, di))
        near = np.where(step_u < 0, np.nextafter(near, 0.0), np.where(step_u > 0, np.nextafter(near, 1.0), near))
        u_inj = np.where(rng.random((ki, di)) < 0.5, near, u_inj)
    edge = rng.random((ki, di))
    u_inj = np.clip(np.where(edge < 0.03, 0.0, np.where(edge > 0.97, 1.0, u_inj)), 0.0, 1.0 - 2.0 ** -53)
    inj = m._dev.sample_injected(u_inj, sym_stride=m._flat.max_segments)
    m._dev.check()
    inj_s = m._flat.decode_entries(inj["syms"].cpu().numpy(), inj["len"].cpu().numpy())
    inj_x = inj["ml2p"].cpu().numpy()
    ok3 = True
    for i in range(ki):
        it = iter(u_inj[i].tolist())
        x, t = o.sample1(lambda: next(it))
        ok3 = ok3 and bits([x])[0] == bits([inj_x[i]])[0] and t == inj_s[i]
    ok = ok and ok3
    uniq = list(dict.fromkeys(want_s))[:1500]
    alphabet = "abcjABF019!@._~ é"
    mut = []
    for t in uniq[:700]:
        if t:
            j = int(rng.integers(0, len(t)))
            mut.append(t[:j] + alphabet[int(rng.integers(0, len(alphabet)))] + t[j + (int(rng.integers(0, 2))):])
    pw = uniq + mut + ["", "~~~~", "a" * 40, "aB1!aB1!"]
    got, _parts = m.score_batch(pw)
    ok2 = np.array_equal(bits(got), bits([o.calc_ml2p(t) for t in pw]))
    return ok and ok2, (f"case {c:3d} bpe-pcfg structures={ns:2d}                 samples={count:5d} "
                        f"sample={'ok' if ok else 'BAD'} injected={'ok' if ok3 else 'BAD'} score={'ok' if ok2 else 'BAD'}")


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        mode = int(rng.integers(0, 5))
        if mode >= 2:
            ok, text = (subword_case, v2_case, bpe_case)[mode - 2](rng, c)
            bad += 0 if ok else 1
            print(text, flush=True)
            continue
        lines, alpha = corpus(rng)
        if int(rng.integers(0, 2)):
            thr = int(rng.choice([1, 2, 3, 5, 10, 30]))
            what = f"backwords thr={thr}"
            try:
                probs, _ = orc.train_backwords(lines, "", "\x00", "\x03", 0, 1, thr, 256)
            except ValueError:  # (no line length reaches the threshold: the reference's trainer raises there, max() of nothing)
                try:
                    BackWordsMonteCarlo(synth.as_file(lines), splitter="", threshold=thr)
                    print(f"case {c:3d} {what}: the oracle's trainer refused the corpus, the package's did not: BAD", flush=True)
                    bad += 1
                except Exception as e:  # noqa: BLE001
                    print(f"case {c:3d} {what}: refused by both trainers ({type(e).__name__})", flush=True)
                continue
            m = BackWordsMonteCarlo(synth.as_file(lines), splitter="", threshold=thr)
            cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_BACKOFF, start_units=["\x00"])
        else:
            n = int(rng.integers(2, 7))
            what = f"nwords n={n}"
            m = NWordsMonteCarlo(synth.as_file(lines), n=n, splitter="")
            probs, _ = orc.train_nwords(lines, n, "", "\x03")
            cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_NGRAM, order=n - 1, start_units=["\x00"] * (n - 1))
        count, sd, first = int(rng.integers(1, 30000)), int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 ** 40))
        res = m._dev.sample(count, sd, first_id=first, max_draws=2000, want_len=True, want_draws=True, sym_stride=20)
        try:
            cnt = m._dev.check()
        except nv.NativeError as e:  # (a model whose walks do not end -- no line of min_len characters, say: NaN on both sides)
            assert e.code == nv.ERR_DIVERGED, e
            cnt = None
        want = cm.sample(count, sd, first_id=first, max_draws=2000, stride=20)
        got_x = res["ml2p"].cpu().numpy()
        lost = np.isnan(want["ml2p"])
        ok = (np.array_equal(np.isnan(got_x), lost) and np.array_equal(bits(got_x[~lost]), bits(want["ml2p"][~lost]))
              and np.array_equal(res["draws"].cpu().numpy().astype(np.uint32), want["draws"])
              and (cnt is None) == bool(lost.any()) and (cnt is None or cnt["draws"] == int(want["draws"].sum())))
        keep = np.flatnonzero(~lost)
        ok = ok and np.array_equal(res["len"].cpu().numpy().astype(np.uint32)[keep], want["len"][keep])
        got_s = m._flat.decode(res["syms"].cpu().numpy()[keep], np.minimum(res["len"].cpu().numpy()[keep], 20))
        want_s = cm.strings({"syms": want["syms"][keep], "len": np.minimum(want["len"][keep], 20)})
        ok = ok and got_s == want_s
        if cnt is None:
            cnt = {"failed": int(lost.sum()), "indirect_draws": -1}
        test = list(dict.fromkeys(lines))[:4000]
        mut = []
        for s in test[:1500]:
            j = int(rng.integers(0, len(s)))
            mut.append(s[:j] + alpha[int(rng.integers(0, len(alpha)))] + s[j + (int(rng.integers(0, 2))):])
        test = test + mut + ["", "\x7f", alpha[0] * 40]
        got, _ = m.score_batch(test)
        ok2 = np.array_equal(bits(np.array(got)), bits(cm.score(test)))
        # parse_file (MonteCarloParent.py:48-69: lines stripped of "\r\n", deduplicated in first-seen order, scored, put
        # in stable order of ml2p) on a text of repeated lines with every kind of line end, against the same loop in
        # Python over the C oracle's scores; the native ingest when the text holds no "\r", the line loop when it does
        import io
        ends = ["\n"] if int(rng.integers(0, 2)) else ["\n", "\r\n", "\r\r\n"]
        picks = rng.integers(0, len(test), size=int(rng.integers(0, 4000)))
        text_f = "".join(test[i] + ends[int(rng.integers(0, len(ends)))] for i in picks)
        if len(picks) and int(rng.integers(0, 2)):
            text_f = text_f.rstrip("\r\n")                                   # (no line end after the last line)
        seen = {}
        for line in io.StringIO(text_f):
            line = line.strip("\r\n")
            seen[line] = seen.get(line, 0) + 1
        uniq_l = list(seen)
        sc_l = cm.score(uniq_l) if uniq_l else np.zeros(0)
        want_p = sorted(((p, seen[p], float(x)) for p, x in zip(uniq_l, sc_l)), key=lambda r: r[2])
        got_p = m.parse_file(io.StringIO(text_f))
        ok4 = len(got_p) == len(want_p) and all(
            (g[0], g[1]) == (w[0], w[1]) and bits([g[2]])[0] == bits([w[2]])[0] for g, w in zip(got_p, want_p))
        ok = ok and ok4
        # injected uniforms AT the rows' thresholds: r = cum[i] / total and its neighbours one ulp below and above (where
        # bisect_right(cum, total * r) changes its answer: ProbLib.py:31-43), 0.0 and the largest double below 1 among
        # random ones -- the draw table decides in r's own domain and must agree with the reference's comparison in
        # total * r's for every one of them
        n_inj, ok6 = injected_check(rng, m, cm, orc.expand_table(probs), 20)
        ok = ok and ok6
        # MonteCarlo.sample(size, sampled_pwds) (MonteCarloParent.py:26-46): the dict of distinct sampled strings built
        # from the GPU's dedup (one row per distinct string reaches the host; strings longer than the kept stride are
        # replayed) against the one built from every sample of the C oracle, content and first-sampled order
        ok5, n_distinct = True, -1
        if not lost.any():
            nd, sd2 = min(count, 5000), int(rng.integers(0, 2 ** 62))
            want_d = cm.sample(nd, sd2, max_draws=1 << 16, stride=64)
            if int(want_d["len"].max()) <= 64 and not np.isnan(want_d["ml2p"]).any():
                ref_d = {}
                for x, pwd in zip(want_d["ml2p"].tolist(), cm.strings({"syms": want_d["syms"], "len": want_d["len"]})):
                    ref_d.setdefault(pwd, [x, 0])[1] += 1
                stride_d = int(rng.choice([32, 7, 3]))
                m._sample_stride = lambda stride_d=stride_d: stride_d
                m.seed(sd2)
                got_d = {"stale": [0.0, 1]}
                keys_d = m.sample(nd, got_d)
                del m._sample_stride
                ok5 = (list(got_d.items()) == list(ref_d.items())
                       and np.array_equal(bits(keys_d.tolist()), bits(want_d["ml2p"])))
                n_distinct = len(ref_d)
        ok = ok and ok5
        # the threshold enumerators (mc_enumerate) against the reference's recursion as the oracle restates it: every
        # string with -log2 p below a random threshold (fewer than 2^threshold of them), in the reference's order
        thr_e = float(rng.uniform(3.0, 15.0))
        if what.startswith("backwords"):
            want_e = orc.enumerate_strings(probs, thr_e, "\x00", "\x03", 4)
            e_ml2p, e_lens, e_syms = m._dev.enumerate(thr_e, 4, 256)
        else:
            want_e = orc.enumerate_strings(probs, thr_e, "\x00", "\x03", 4, order=n - 1)
            e_ml2p, e_lens, e_syms = m._dev.enumerate(thr_e, 4, 30)
        ok3 = (len(e_ml2p) == len(want_e) and m._flat.decode(e_syms, e_lens) == [w[0] for w in want_e]
               and np.array_equal(bits(e_ml2p), bits([w[1] for w in want_e])))
        if not (ok and ok2 and ok3):
            bad += 1
        print(f"case {c:3d} {what:18s} lines={len(lines):5d} alphabet={len(alpha):2d} samples={count:5d} "
              f"failed={cnt['failed']} indirect={cnt['indirect_draws']} sample={'ok' if ok else 'BAD'} "
              f"score={'ok' if ok2 else 'BAD'} parse_file={len(want_p)} {'ok' if ok4 else 'BAD'} dedup={n_distinct} {'ok' if ok5 else 'BAD'} injected={n_inj} {'ok' if ok6 else 'BAD'} enumerated={len(want_e)} {'ok' if ok3 else 'BAD'}", flush=True)
    print("ALL OK" if not bad else f"{bad} FAILED")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
