"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the
C-ABI of libmi355mc.so (montecarlo_amd._native via the host classes):

 (a) per-password -log2 p == reference calc_ml2p (golden, bit-exact; tolerance of the
     north star is 1e-9, the implementation is exact);
 (b) same uniform stream -> bit-identical strings / ml2p / draw counts as the reference
     sampler (golden), in Philox mode and in injected-uniform mode;
 (c) identical sample set -> guess numbers within 1e-9 relative of the reference (golden);
 (d) sample distribution vs the reference's Mersenne-Twister sampler (KS / chi-square),
     plus bit-exact agreement with the C oracle on larger runs.
"""
import io
import os
from math import ceil

import numpy as np
import pytest
import torch

from conftest import fh, golden, golden_lines, hx  # noqa: F401
from montecarlo_amd import _native as nv, synth
from montecarlo_amd.backwords.backwords_secondary_trainer import backwords_counter as secondary_counter
from montecarlo_amd.backwords_secondary_simulator import BackWordsSecondaryMonteCarlo
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.engine import DeviceRank
from montecarlo_amd.lib4mc.MonteCarloLib import MonteCarloLib
from montecarlo_amd.nwords_simulator import NWordsMonteCarlo
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
from oracle import c_oracle
from oracle.philox import uniform53

pytestmark = pytest.mark.gpu
REL = 1e-9      # north-star tolerance for guess numbers


def device_samples(model, fx, n=None, injected=False):
    items = fx["samples"]["items"][:n]
    seed = fx["samples"]["seed"]
    dev = model._dev
    if injected:
        dmax = max(it[2] for it in items)
        u = np.array([[uniform53(seed, i, k) for k in range(dmax)] for i in range(len(items))])
        res = dev.sample_injected(u, sym_stride=300)
    else:
        res = dev.sample(len(items), seed, want_draws=True, sym_stride=300)
    dev.check()
    strings = model._flat.decode(res["syms"].cpu().numpy(), res["len"].cpu().numpy(),
                                 drop=getattr(model, "start_chr", ""))
    got = [[hx(a), s, int(d)] for a, s, d in zip(res["ml2p"].cpu().numpy(), strings, res["draws"].cpu().numpy())]
    assert got == items


def device_scores(model, fx):
    pw = [s[0] for s in fx["scores"]]
    ml2p, parts = model.score_batch(pw)
    assert [hx(x) for x in ml2p] == [s[1] for s in fx["scores"]]
    assert parts == [s[2] for s in fx["scores"]]


@pytest.mark.parametrize("n", [2, 4])
def test_nwords_char(train_char, n):
    fx = golden(f"nwords_n{n}_char.json")
    m = NWordsMonteCarlo(synth.as_file(train_char), n=n, splitter="")
    device_samples(m, fx)
    device_samples(m, fx, n=200, injected=True)
    device_scores(m, fx)


@pytest.mark.parametrize("thr", [2, 10])
def test_backwords_char(train_char, thr):
    fx = golden(f"backwords_thr{thr}_char.json")
    m = BackWordsMonteCarlo(synth.as_file(train_char), splitter="", threshold=thr)
    device_samples(m, fx)
    device_samples(m, fx, n=200, injected=True)
    device_scores(m, fx)


def test_subword_sampling(train_tab):
    m = NWordsMonteCarlo(synth.as_file(train_tab), n=3, splitter="\t")
    device_samples(m, golden("nwords_n3_tab.json"))
    m = BackWordsMonteCarlo(synth.as_file(train_tab), splitter="\t", threshold=3)
    device_samples(m, golden("backwords_thr3_tab.json"))


def test_subword_scoring(train_tab, train_char):
    """calc_ml2p for sub-word alphabets: the DFS kernel (every segmentation, minimum, iter_max)
    against the reference's values and components (golden), through the plugin surface."""
    m = NWordsMonteCarlo(synth.as_file(train_tab), n=3, splitter="\t")
    device_scores(m, golden("nwords_n3_tab.json"))
    m = BackWordsMonteCarlo(synth.as_file(train_tab), splitter="\t", threshold=3)
    fx = golden("backwords_thr3_tab.json")
    device_scores(m, fx)
    assert m.calc_ml2p(fx["scores"][0][0]) == (fh(fx["scores"][0][1]), fx["scores"][0][2])
    scored = m.parse_file(synth.as_file(golden_lines("test_tab.txt")), using_component=True)
    want = {s[0]: (s[1], s[2]) for s in fx["scores"]}
    assert all(isinstance(c, list) for c, _, _ in scored)
    assert [x for _, _, x in scored] == sorted(x for _, _, x in scored)
    for comps, _cnt, x in scored:
        pwd = "".join(comps)
        if pwd in want:
            assert (hx(x), comps) == want[pwd]
    m2 = BackWordsMonteCarlo(synth.as_file(train_tab), splitter="\t", threshold=3, max_iter=2)
    device_scores(m2, golden("backwords_thr3_tab_maxiter2.json"))
    # the DFS kernel on a single-character alphabet agrees with the lookup-chain kernel and the golden values
    mc = BackWordsMonteCarlo(synth.as_file(train_char), splitter="", threshold=10)
    fxc = golden("backwords_thr10_char.json")
    pw = [s[0] for s in fxc["scores"]]
    ml2p, parts = mc.score_batch_dfs(pw)
    assert [hx(x) for x in ml2p] == [s[1] for s in fxc["scores"]]
    assert parts == [s[2] for s in fxc["scores"]]
    # a password longer than the declared bound is reported, not silently mis-scored
    cps, off, max_len = mc._flat.encode_text(["abc", "x" * 40])
    res, _ = mc._dev.score_words(cps, off, 10, 10 ** 35)
    with pytest.raises(Exception, match="longer than the max_len"):
        mc._dev.check()
    assert np.isnan(res.cpu().numpy()[1])


def test_bpe_pcfg():
    """BpePcfgSim on the GPU: samples (Philox and injected uniforms) and calc_ml2p against the
    reference's recorded values; the plugin surface end to end (sample -> MonteCarloLib ->
    parse_file -> ml2p_iter2gc), which the reference itself cannot run."""
    import os
    from conftest import GOLDEN
    from montecarlo_amd.bpe_simulator import BpePcfgSim
    fx = golden("bpe.json")
    m = BpePcfgSim(os.path.join(GOLDEN, "bpe_model"))
    items, seed = fx["samples"]["items"], fx["samples"]["seed"]
    for injected in (False, True):
        if injected:
            dmax = max(it[2] for it in items)
            u = np.array([[uniform53(seed, i, k) for k in range(dmax)] for i in range(len(items))])
            res = m._dev.sample_injected(u, sym_stride=m._flat.max_segments)
        else:
            res = m._dev.sample(len(items), seed, want_draws=True, want_len=True, sym_stride=m._flat.max_segments)
        m._dev.check()
        strings = m._flat.decode_entries(res["syms"].cpu().numpy(), res["len"].cpu().numpy())
        got = [[hx(a), s, int(d)] for a, s, d in zip(res["ml2p"].cpu().numpy(), strings, res["draws"].cpu().numpy())]
        assert got == items
    pw = [s[0] for s in fx["scores"]] + ["", "zzzzzzzzzzzzzzzzzzzzzzzzzzzzzz"]
    ml2p, parts = m.score_batch(pw)
    assert [hx(x) for x in ml2p[:-2]] == [s[1] for s in fx["scores"]]
    assert ml2p[-2:] == [1022.0, 1022.0] and parts[0] == [pw[0]]
    assert m.calc_ml2p(pw[0]) == (fh(fx["scores"][0][1]), [pw[0]])
    m.seed(seed)
    seen = {}
    keys = m.sample(len(items), seen)
    assert [hx(x) for x in keys] == [it[0] for it in items]
    assert list(seen)[:3] == list(dict.fromkeys(it[1] for it in items))[:3]
    mc = MonteCarloLib(keys)
    gc = mc.ml2p_iter2gc(m.parse_file(synth.as_file([s[0] for s in fx["scores"]])), need_resort=True)
    assert len(gc) == len(fx["scores"]) and all(a[3] < b[3] for a, b in zip(gc, gc[1:]))


def test_secondary_and_v2(train_char, train_v2):
    fx = golden("secondary_thr4_char.json")
    cfg = {"start_chr": "\x03", "end_chr": "\x00", "threshold": 4}
    counts, words = secondary_counter(train_char[:2500], "", "\x03", "\x00", 0, 1, 256, 4)
    m = BackWordsSecondaryMonteCarlo((counts, words, cfg))
    device_samples(m, fx["round0"])
    device_scores(m, fx["round0"])
    counts, words = secondary_counter(train_char[2500:3200], "", "\x03", "\x00", 0, 1, 256, 4, nwords_dict=counts,
                                      words=words)
    m = BackWordsSecondaryMonteCarlo((counts, words, cfg))
    device_samples(m, fx["round1"])
    device_scores(m, fx["round1"])
    m = NWords2MonteCarlo(synth.as_file(train_v2), 4)
    device_samples(m, golden("v2_n4.json"))


def test_secondary_main_matches_reference_run(tmp_path):
    """BASELINE config 5: `backwords_secondary_main --strategy samples 2` end to end (count merge ->
    model -> sample with strings -> rank -> score -> cracked set -> next round) against the output
    files of the reference's own run with the same streams (golden/secondary_main.json)."""
    import os
    import random
    from conftest import GOLDEN
    from montecarlo_amd import backwords_secondary_main as main
    fx = golden("secondary_main.json")
    save = str(tmp_path / "save")
    random.seed(fx["meta"]["seed"])
    main.wrapper(["-i", os.path.join(GOLDEN, fx["meta"]["train"]), "-t", os.path.join(GOLDEN, fx["meta"]["test"]),
                  "-s", save] + fx["meta"]["args"])
    from secondary_check import check_saved
    check_saved(save, fx)


@pytest.mark.parametrize("strategy", ["guesses", "hits", "auto_hits"])
def test_secondary_main_other_strategies_match_reference_run(tmp_path, strategy):
    """`--strategy guesses <g1> <g2>`, `hits <h1> <h2>` and `auto_hits <factor> <base> <termination>` of the
    config-5 driver (reference backwords_secondary_main.py:139-163,70): every output file against the
    reference's own run of the same command (golden/secondary_strategies.json)."""
    import os
    import random
    from conftest import GOLDEN
    from montecarlo_amd import backwords_secondary_main as main
    fx = golden("secondary_strategies.json")[strategy]
    save = str(tmp_path / "save")
    random.seed(fx["meta"]["seed"])
    main.wrapper(["-i", os.path.join(GOLDEN, fx["meta"]["train"]), "-t", os.path.join(GOLDEN, fx["meta"]["test"]),
                  "-s", save] + fx["meta"]["args"])
    from secondary_check import check_saved
    check_saved(save, fx)


@pytest.mark.timeout(170)
def test_secondary_main_sharded_over_two_ranks(tmp_path):
    """Config 5 with the sample set sharded over ranks: `backwords_secondary_main --strategy samples 2` under
    torch.distributed.run with 2 ranks (they share this box's GPU and talk over gloo; on a multi-GPU node the
    same code runs one rank per GPU over RCCL).  The product sampler draws each rank's share of the global
    sample ids (`MonteCarlo.sample` under `dist.use_group`), the `sampled_pwds` dicts are merged, `MonteCarloLib`
    all-reduces its queries: rank 0's files must equal the reference's single-process run."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, TQDM_DISABLE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "scripts", "rehearse_secondary.py")]
    done = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=150)
    assert done.returncode == 0, done.stdout[-2000:] + done.stderr[-4000:]
    assert "rehearse_secondary ok: 2 rank(s)" in done.stdout


def test_plugin_surface_matches_reference_run(train_char):
    """sample -> MonteCarloLib -> parse_file -> ml2p_iter2gc -> write2, the whole driver of
    backwords_simulator.py:90-94, against the reference's recorded run (rank.json)."""
    fx = golden("rank.json")
    m = BackWordsMonteCarlo(synth.as_file(train_char), splitter="", threshold=10)
    m.seed(fx["meta"]["seed"])
    keys = m.sample(fx["meta"]["n"])
    assert [hx(x) for x in keys] == fx["draw_order"]
    mc = MonteCarloLib(keys)
    assert [hx(x) for x in keys] == fx["sorted"]                       # sorted in place, like the reference
    pos = mc.device_rank.positions.cpu().numpy()
    want = np.array([fh(x) for x in fx["positions"]])
    assert np.all(np.abs(pos - want) <= REL * want)
    for x, w in fx["probes"]:
        got = mc.ml2p2rank(fh(x))
        assert abs(float(got) - fh(w)) <= REL * fh(w)
    scored = m.parse_file(synth.as_file(golden_lines("test_char.txt")))
    assert [[p, c, hx(x)] for p, c, x in scored] == fx["scored"]
    gc = mc.ml2p_iter2gc(scored)
    for (p, x, c, r, k, ratio), (wp, wx, wc, wr, wk, wratio) in zip(gc, fx["gc"]):
        assert (p, hx(x), c, k, hx(ratio)) == (wp, wx, wc, wk, wratio)
        assert abs(r - int(wr)) <= REL * int(wr) + 1e-12, (p, r, wr)
    exact = sum(1 for g, w in zip(gc, fx["gc"]) if g[3] == int(w[3]))
    assert exact >= len(gc) - 2                                        # ceil() may flip on an exact integer
    buf = io.StringIO()
    mc.write2(buf)
    # every line of write2 against the reference's text; a line whose rank flipped (<= 2 of them, above) is
    # compared with its rank field taken out
    got_lines, want_lines = buf.getvalue().split("\n"), fx["write2"].split("\n")
    assert len(got_lines) == len(want_lines) == len(gc) + 1 and got_lines[-1] == want_lines[-1] == ""
    for g, w, a, b in zip(gc, fx["gc"], got_lines, want_lines):
        if g[3] == int(w[3]):
            assert a == b
        else:
            fa, fb = a.split("\t"), b.split("\t")
            assert fa[:3] + fa[4:] == fb[:3] + fb[4:] and fa[3] == str(g[3])
    gc0 = mc.ml2p_iter2gc(scored, add1=False)
    assert sum(1 for g, w in zip(gc0, fx["gc_add0_ranks"]) if g[3] == int(w)) >= len(gc0) - 2
    # list input: uploaded, sorted in place on the device, same numbers
    lst = [fh(x) for x in fx["draw_order"]]
    mc2 = MonteCarloLib(lst)
    assert [hx(x) for x in lst] == fx["sorted"]
    assert mc2.ml2p2rank(33.3) == mc.ml2p2rank(33.3)


# ---------------------------------------------------------------------------------- larger runs
@pytest.fixture(scope="module")
def big_model():
    train = synth.zipf_corpus(100000)
    m = BackWordsMonteCarlo(synth.as_file(train), splitter="", threshold=10)
    cm = c_oracle.COracleModel(m.nwords, c_oracle.KIND_BACKOFF, start_units=["\x00"])
    return m, cm


def test_sampling_matches_c_oracle(big_model):
    m, cm = big_model
    n, seed, first = 200000, 0xC0FFEE, 12345678901
    res = m._dev.sample(n, seed, first_id=first, want_len=True, want_draws=True, sym_stride=24)
    cnt = m._dev.check()
    want = cm.sample(n, seed, first_id=first, stride=24)
    assert np.array_equal(res["ml2p"].cpu().numpy().view(np.uint64), want["ml2p"].view(np.uint64))
    assert np.array_equal(res["draws"].cpu().numpy().astype(np.uint32), want["draws"])
    assert np.array_equal(res["len"].cpu().numpy().astype(np.uint32), want["len"])
    assert cnt["draws"] == int(want["draws"].sum()) and cnt["failed"] == 0
    got_s = m._flat.decode(res["syms"].cpu().numpy(), np.minimum(res["len"].cpu().numpy(), 24))
    want_s = cm.strings({"syms": want["syms"], "len": np.minimum(want["len"], 24)})
    assert got_s == want_s


def test_launch_geometry_does_not_change_results(big_model):
    m, _ = big_model
    a = m._dev.sample(50000, 7, first_id=1000)["ml2p"]
    b = torch.cat([m._dev.sample(20000, 7, first_id=1000)["ml2p"], m._dev.sample(30000, 7, first_id=21000)["ml2p"]])
    m._dev.check()
    assert torch.equal(a.view(torch.int64), b.view(torch.int64))
    # mc_sample_tune: workgroups per CU, skew of their ranges, KB of the draw table in LDS
    want = m._dev.sample(1_200_007, 9, first_id=2 ** 33, want_len=True, want_draws=True, sym_stride=12)
    try:
        for geometry in ((7, 0.55, 15), (7, 0.0, 0), (3, 0.9, 40), (20, 0.25, 8), (1, -1.0, -1)):
            m._dev.tune(*geometry)
            got = m._dev.sample(1_200_007, 9, first_id=2 ** 33, want_len=True, want_draws=True, sym_stride=12)
            assert torch.equal(got["ml2p"].view(torch.int64), want["ml2p"].view(torch.int64)), geometry
            for k in ("len", "draws", "syms"):
                assert torch.equal(got[k], want[k]), (geometry, k)
        with pytest.raises(RuntimeError, match="mc_sample_tune"):
            m._dev.tune(65)
    finally:
        m._dev.tune()
    m._dev.check()


def test_sample_carry_copies_beside_the_same_samples(big_model):
    """mc_sample_carry (include/mi355mc.h): the samples are mc_sample's bit for bit and the copy taken along arrives
    whole -- in pinned host memory and in device memory, sizes around the copy workgroups' strides, size 0, a count
    of 0 (the runtime's copy then); misaligned sizes are refused before anything is launched."""
    m, _ = big_model
    dm = m._dev
    want = dm.sample(70000, 11, first_id=5)["ml2p"]
    g = torch.Generator().manual_seed(5)
    for words in (2, 8 * 256 * 2 - 2, 8 * 256 * 2 * 8 * 3 + 2, 1_000_002):           # int64 words, 16 bytes a pair
        src = torch.randint(-2 ** 62, 2 ** 62, (words,), generator=g, dtype=torch.int64).cuda()
        for dst in (torch.zeros(words, dtype=torch.int64).pin_memory(), torch.zeros(words, dtype=torch.int64, device="cuda")):
            got = dm.sample(70000, 11, first_id=5, carry=(src, dst))["ml2p"]
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.int64), want.view(torch.int64))
            assert torch.equal(dst.cpu(), src.cpu())
    src = torch.arange(64, dtype=torch.int64).cuda()
    dst = torch.zeros(64, dtype=torch.int64).pin_memory()
    dm.sample(0, 11, carry=(src, dst))
    torch.cuda.synchronize()
    assert torch.equal(dst, src.cpu())
    dm.check()
    with pytest.raises(RuntimeError, match="multiple of 16"):
        dm.sample(100, 11, carry=(src[:3], dst[:3]))
    with pytest.raises(RuntimeError, match="aligned"):
        dm.sample(100, 11, carry=(src[1:5], dst[1:5]))


def test_sampled_pwds_dedup_on_gpu(big_model):
    """MonteCarlo.sample(size, sampled_pwds): the dict built from the GPU-side dedup (one row per
    distinct string reaches the host) equals the one built by decoding every sample, in content and
    in first-sampled order -- also when most strings are longer than the kept stride (replays)."""
    m, cm = big_model
    n, seed = 60000, 31337
    want = cm.sample(n, seed, stride=40)
    strings = cm.strings({"syms": want["syms"], "len": want["len"]})
    ref = {}
    for x, pwd in zip(want["ml2p"].tolist(), strings):
        ref.setdefault(pwd, [x, 0])[1] += 1
    for stride in (32, 7):
        m._sample_stride = lambda stride=stride: stride
        m.seed(seed)
        got = {"stale": [0.0, 1]}
        keys = m.sample(n, got)
        assert list(got.items()) == list(ref.items())
        assert np.array_equal(np.array(keys.tolist()).view(np.uint64), want["ml2p"].view(np.uint64))
    del m._sample_stride
    m.seed(seed)
    more = dict(got)
    m.sample(1000, more, clearIfNotNone=False)
    assert len(more) >= len(got) and list(more)[:len(got)] == list(got)


def test_parse_file_native_ingest_equals_line_loop(big_model):
    """parse_file through mc_lines_ingest == the reference's line loop (lib4mc/MonteCarloParent.py:56-68;
    a text with '\\r' line ends takes the loop), 1e5 lines with duplicates, empty and out-of-model lines."""
    m, _ = big_model
    lines = synth.zipf_corpus(100000, oov_frac=0.02, stream=3) + ["", "", "a" * 300, "pässwörd"]
    fast = m.parse_file(io.StringIO("\n".join(lines) + "\n"))
    loop = m.parse_file(io.StringIO("\r\n".join(lines) + "\r\n"))
    assert fast == loop
    assert sum(c for _, c, _ in fast) == len(lines)
    assert [x for _, _, x in fast] == sorted(x for _, _, x in fast)
    comp_fast = m.parse_file(io.StringIO("\n".join(lines[:5000])), using_component=True)
    comp_loop = m.parse_file(io.StringIO("\r\n".join(lines[:5000])), using_component=True)
    assert comp_fast == comp_loop


def test_gc_columns_and_native_write2(big_model):
    """ml2p_iter2gc's vectorised columns == the reference's running loop (lib4mc/MonteCarloLib.py:47-58), and
    write2 through mc_format_gc == its f-string lines (:65-68), on 2e4 distinct test passwords; a list the
    caller changed before write2 is written as changed."""
    m, _ = big_model
    m.seed(5)
    mc = MonteCarloLib(m.sample(200000))
    test = synth.zipf_corpus(60000, oov_frac=0.02, stream=7)
    scored = m.parse_file(synth.as_file(test))
    assert len(scored) > 5000
    gc = mc.ml2p_iter2gc(scored)
    total, run, prev = sum(c for _, c, _ in scored), 0, 0
    for (pwd, x, cnt, rank, cracked, ratio), (wp, wc, wx) in zip(gc, scored):
        run += wc
        assert (pwd, x, cnt, cracked, ratio) == (wp, wx, wc, run, run / total * 100)
        assert rank > prev
        prev = rank
    want = "".join(f"{p}\t{x:.8f}\t{c}\t{r}\t{k}\t{q:5.2f}\n" for p, x, c, r, k, q in gc)
    buf = io.StringIO()
    mc.write2(buf)
    assert buf.getvalue() == want
    gc = mc.ml2p_iter2gc(scored)
    gc[0] = ("changed", 1.0, 2, 3, 4, 5.0)
    buf = io.StringIO()
    mc.write2(buf)
    assert buf.getvalue().split("\n")[0] == "changed\t1.00000000\t2\t3\t4\t 5.00"


def test_driver_flow_edge_sizes(big_model):
    """Empty, one-line, two-equal-lines and just-over-the-native-writer's-threshold test files through
    parse_file -> ml2p_iter2gc -> write2: the native-ingest route and the reference's line loop (taken for a text
    with '\\r') give the same lists and the same bytes."""
    m, _ = big_model
    m.seed(13)
    mc = MonteCarloLib(m.sample(50000))
    words = synth.zipf_corpus(700, oov_frac=0.05, stream=5)
    for lines in ([], ["password1"], ["abc123", "abc123"], [""], words[:3], words[:513], words):
        outs = []
        for sep in ("\n", "\r\n"):
            text = sep.join(lines) + (sep if lines else "")
            scored = m.parse_file(io.StringIO(text, newline=""))
            gc = mc.ml2p_iter2gc(scored)
            buf = io.StringIO()
            mc.write2(buf)
            outs.append((list(scored), list(gc), buf.getvalue()))
        assert outs[0] == outs[1], len(lines)
        assert sum(c for _, c, _ in outs[0][0]) == len(lines)
        assert outs[0][2].count("\n") == len(set(lines))


def test_c_abi_collective_on_one_rank():
    """mc_comm_unique_id / mc_comm_init / mc_allreduce_rank / mc_comm_destroy (RCCL bound at run time) with a
    world of one rank -- what a 1-GPU box can run: the sum over one rank leaves positions and counts as they
    are, twice in a row, on the caller's stream.  (More ranks need more GPUs: RCCL refuses two ranks on one
    device; the sum itself is covered over gloo in tests/test_dist_gloo.py.)"""
    from montecarlo_amd.dist import NativeComm
    comm = NativeComm(NativeComm.unique_id(), 0, 1)
    g = torch.Generator().manual_seed(2)
    pos = (torch.rand(100001, generator=g, dtype=torch.float64) * 1e12).cuda()
    cnt = torch.randint(0, 1 << 40, (100001,), generator=g, dtype=torch.int64).cuda()
    want_p, want_c = pos.clone(), cnt.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        comm.combine(pos, cnt)
        comm.combine(pos)
    side.synchronize()
    assert torch.equal(pos, want_p) and torch.equal(cnt, want_c)
    comm.close()


def test_order_keys_is_pythons_stable_sort():
    """mc_order_keys == sorted(range(t), key=x.__getitem__) (lib4mc/MonteCarloParent.py:68): ties keep their
    input order, -0.0 ties with 0.0, unparseable scores (1022.0) and inf sort where Python puts them."""
    from montecarlo_amd.engine import device_order
    rng = np.random.default_rng(21)
    cases = {
        "ties": np.round(rng.random(300000) * 40, 1),
        "distinct": rng.random(200001) * 90,
        "unparseable": np.where(rng.random(100000) < 0.3, 1022.0, rng.random(100000) * 60),
        "zeros": np.where(rng.random(5000) < 0.5, -0.0, 0.0) * np.where(rng.random(5000) < 0.2, 0.0, 1.0),
        "signed": np.concatenate([rng.standard_normal(70000) * 30, [np.inf, -np.inf, 0.0, -0.0, np.inf, 1022.0]]),
        "one": np.array([3.5]),
        "equal": np.full(70000, 17.25),
    }
    for name, x in cases.items():
        x = np.ascontiguousarray(x, dtype=np.float64)
        srt, order = device_order(torch.from_numpy(x).cuda())
        order = order.cpu().numpy().view(np.uint32)
        want = np.argsort(x, kind="stable")                  # (numpy compares -0.0 == 0.0 as Python does)
        assert np.array_equal(order, want.astype(np.uint32)), name
        assert np.array_equal(srt.cpu().numpy().view(np.uint64), x[want].view(np.uint64)), name
    srt, order = device_order(torch.empty(0, dtype=torch.float64, device="cuda"))
    assert srt.numel() == 0 and order.numel() == 0
    # above 2^24 keys the second half is a key sort of (L << 32 | index) words instead of the LSD passes
    g = torch.Generator().manual_seed(5)
    x = (torch.randint(0, 3_000_000, (17_000_001,), generator=g).double() * 2.0 ** -10).cuda()
    srt, order = device_order(x)
    want = torch.sort(x, stable=True)
    assert torch.equal(order.long() & 0xFFFFFFFF, want[1]) and torch.equal(srt, want[0])


def test_order_keys_places_groups_of_every_size():
    """Passwords that share a score keep their input order whatever the size of the group: test sets made of groups
    of 2, up to 8, up to 64, up to 512 and thousands of equal scores, Zipf-sized groups like the bench's test set
    (of its 1e6 passwords 145,000 are one password), groups larger than 2^18, tiny inputs that are one group, and one
    workspace serving several calls.  (Written for a second half of mc_order_keys that placed the groups by counting
    and then put each in order by one of four routes; it was correct and slower than the stable LSD passes on both
    unique and heavily repeated test sets -- DESIGN.md 3.6 -- so the passes stay, with these cases.)"""
    from montecarlo_amd.engine import device_order
    rng = np.random.default_rng(77)

    def groups(sizes):
        vals = rng.permutation(len(sizes)).astype(np.float64) * 0.37 + 1.0
        x = np.repeat(vals, sizes)
        return x[rng.permutation(x.size)]

    zipf = np.minimum(rng.zipf(1.3, 20000), 150000)
    cases = {
        "pairs": groups(np.full(40000, 2)),
        "up to 8": groups(rng.integers(1, 9, 30000)),
        "9 to 64": groups(rng.integers(9, 65, 4000)),
        "65 to 512": groups(rng.integers(65, 513, 600)),
        "513 and more": groups(rng.integers(513, 5000, 90)),
        "mixed": groups(np.concatenate([np.full(50000, 1), rng.integers(2, 9, 5000), rng.integers(9, 65, 800),
                                        rng.integers(65, 513, 100), [513, 30000, 144633]])),
        "zipf": groups(zipf[: np.searchsorted(np.cumsum(zipf), 900000)]),
        "three windows": groups(np.array([300000, 280000, 1, 2, 9, 70, 600, 250000])),
        "nine": np.full(9, 2.5),
        "600": np.full(600, 2.5),
        "513 then one": np.concatenate([np.full(513, 7.0), [3.0]]),
        "2^18 + 7 equal": np.full((1 << 18) + 7, 1.5),
    }
    for name, x in cases.items():
        x = np.ascontiguousarray(x, dtype=np.float64)
        srt, order = device_order(torch.from_numpy(x).cuda())
        want = np.argsort(x, kind="stable")
        assert np.array_equal(order.cpu().numpy().view(np.uint32), want.astype(np.uint32)), name
        assert np.array_equal(srt.cpu().numpy().view(np.uint64), x[want].view(np.uint64)), name
    # the same workspace again with other keys
    ws = None
    for seed in (1, 2, 3):
        x = torch.from_numpy(np.round(np.random.default_rng(seed).random(250000) * 300, 0)).cuda()
        if ws is None:
            from montecarlo_amd import _native as nv
            ws = torch.empty(nv.lib.mc_order_workspace_bytes(x.numel()), dtype=torch.uint8, device="cuda")
        srt, order = device_order(x, workspace=ws)
        want = torch.sort(x, stable=True)
        assert torch.equal(order.long() & 0xFFFFFFFF, want[1]) and torch.equal(srt, want[0])


def test_order_in_two_halves_equals_the_whole():
    """mc_order_keys_begin + mc_order_keys_end == mc_order_keys (same sorted bits, same order), and what the first
    half leaves for the queries -- the scores ascending as numbers -- equals the sorted scores numerically (-0.0
    comes back as 0.0: the same number to a bisection), also when the halves run on two streams."""
    from montecarlo_amd import _native as nv
    from montecarlo_amd.engine import device_order, device_order_begin, device_order_end
    rng = np.random.default_rng(5)
    cases = {
        "scores": np.round(rng.random(400000) * 60, 2),
        "zeros and signs": np.concatenate([np.where(rng.random(3000) < 0.5, -0.0, 0.0), rng.standard_normal(5000), [np.inf, 1022.0]]),
        "few": np.array([2.0, 1.0, 2.0]),
        "above 2^24": None,
    }
    for name, x in cases.items():
        if x is None:
            g = torch.Generator().manual_seed(8)
            xd = (torch.randint(0, 1_000_000, (16_800_000,), generator=g).double() * 2.0 ** -8).cuda()
        else:
            xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda()
        t = xd.numel()
        want_sorted, want_order = device_order(xd)
        ws = torch.empty(nv.lib.mc_order_workspace_bytes(t), dtype=torch.uint8, device="cuda")
        srt, vals = torch.empty_like(xd), torch.empty_like(xd)
        order = torch.empty(t, dtype=torch.int32, device="cuda")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            device_order_begin(xd, srt, vals, ws)
            half = torch.cuda.Event()
            half.record()
            device_order_end(xd, srt, order, ws)
        torch.cuda.current_stream().wait_event(half)
        seen = vals.clone()                                  # (what a query launched now would read)
        side.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(order, want_order), name
        assert torch.equal(srt.view(torch.int64), want_sorted.view(torch.int64)), name
        assert torch.equal(seen, want_sorted) and torch.equal(vals, want_sorted), name        # (numerically: -0.0 == 0.0)
        assert not bool((vals.view(torch.int64) == -(1 << 63)).any()), name                  # no -0.0 among the values


def test_guess_numbers_equal_reference_loop():
    """mc_guess_numbers == the reference's running expression rank = ceil(max(estimate, prev + add))
    (lib4mc/MonteCarloLib.py:47-55) on ascending estimates, across chunk borders, with estimates that do not
    fit int64 (the kernel reports where they start) and with the sharded form (cnt == NULL)."""
    from math import ceil
    from montecarlo_amd.engine import device_guess_numbers
    rng = np.random.default_rng(3)
    for t, scale in ((3000, 1e3), (300000, 1e9), (50000, 2.0 ** 61.9), (3000, 2.0 ** 70), (2049, 5.0), (1, 9.0)):
        est = np.sort(rng.random(t)) * scale
        est[:5] = [1.0, 1.0, 1.0, 2.0, 2.5][:min(5, t)]
        cnt = (rng.random(t) > 0.05).astype(np.int64)
        pos = np.where(cnt > 0, est, 0.0)
        for add1 in (True, False):
            want, prev = [], 0
            for e, k in zip(pos.tolist(), cnt.tolist()):
                prev = ceil(max(e if k > 0 else 1, prev + (1 if add1 else 0)))
                want.append(prev)
            for with_cnt in (True, False):
                ranks, exact, _ = device_guess_numbers(torch.from_numpy(pos).cuda(),
                                                       torch.from_numpy(cnt).cuda() if with_cnt else None, add1)
                exact = int(exact.item())
                fits = next((i for i, e in enumerate(pos.tolist()) if ceil(e) >= 2 ** 62), t)
                assert exact == fits, (t, scale, add1)
                got = ranks.cpu().numpy()
                assert got[:exact].tolist() == want[:exact], (t, scale, add1, with_cnt)
                # beyond the exact prefix the array carries the estimates (fp64 bits) for the caller's integers
                assert np.array_equal(got[exact:].view(np.float64), np.where(cnt > 0, pos, 1.0)[exact:])
    ranks, exact, _ = device_guess_numbers(torch.empty(0, dtype=torch.float64, device="cuda"), None, True)
    assert int(exact.item()) == 0
    # a NaN position (a failed index, or one shard's in the sum of all shards) is never read as "no sample at or
    # below": the exact prefix ends there, with and without counts
    pos = torch.tensor([3.0, 5.0, float("nan"), 9.0], dtype=torch.float64, device="cuda")
    for cnt in (None, torch.ones(4, dtype=torch.int64, device="cuda")):
        ranks, exact, _ = device_guess_numbers(pos, cnt, True)
        assert int(exact.item()) == 2 and ranks[:2].tolist() == [3, 5]


def test_driver_flow_on_columns_equals_tuple_flow(big_model):
    """parse_file -> ml2p_iter2gc -> write2 as the drivers chain them (backwords_simulator.py:90-94) never builds
    a tuple here (montecarlo_amd.columns); the same flow fed with plain lists of tuples -- which takes the
    reference's per-row route through the same entry points -- must write the same bytes, also when guess
    numbers leave int64 (synthetic sample keys up to 120 bits)."""
    m, _ = big_model
    m.seed(9)
    g = torch.Generator().manual_seed(4)
    few = MonteCarloLib((torch.rand(5000, generator=g, dtype=torch.float64) * 120).cuda())   # positions reach 2^107
    many = MonteCarloLib(m.sample(300000))
    test = synth.zipf_corpus(40000, oov_frac=0.02, stream=11) + ["zq" * 40, "p\u00e4ss", "", "\U0001f600x"]
    for mc, add1 in ((many, True), (many, False), (few, True)):
        scored = m.parse_file(synth.as_file(test))
        assert type(scored).__name__ == "ScoredList" and scored.pristine
        gcl = mc.ml2p_iter2gc(scored, add1=add1)
        assert gcl._rows is None and scored._rows is None            # nobody looked: still columns
        a = io.StringIO()
        mc.write2(a)
        rows = list(m.parse_file(synth.as_file(test)))
        gc2 = mc.ml2p_iter2gc(rows, add1=add1)
        assert type(gc2[0]) is tuple
        prev = 0
        for (_, x, _, rank, _, _) in gc2:
            est = mc.ml2p2rank(x) if rank != prev + (1 if add1 else 0) else None
            assert est is None or rank == ceil(max(est, prev + (1 if add1 else 0)))
            prev = rank
        b = io.StringIO()
        mc.write2(b)
        want = "".join(f"{p}\t{x:.8f}\t{c}\t{r}\t{k}\t{q:5.2f}\n" for p, x, c, r, k, q in gc2)
        assert a.getvalue() == b.getvalue() == want
    assert max(r[3] for r in gc2) > 2 ** 63                           # the Python-int suffix was exercised


def test_scoring_matches_c_oracle(big_model):
    m, cm = big_model
    test = list(dict.fromkeys(synth.zipf_corpus(30000, oov_frac=0.02, stream=1))) + ["", "a" * 300]
    got, _ = m.score_batch(test)
    want = cm.score(test)
    assert np.array_equal(np.array(got).view(np.uint64), want.view(np.uint64))
    assert 0 < int((want == 1022.0).sum()) < len(test)


def test_scoring_hash_path_equals_dense_path(big_model):
    """k_score<DENSE> (one gather per step) and k_score<hash> (probe + backoff chain) agree."""
    from montecarlo_amd import _native as nv
    from montecarlo_amd.engine import DeviceModel
    from montecarlo_amd.tables import FlatTable
    m, cm = big_model
    assert len(m._flat.array(nv.ARR_DENSE)) > 0
    flat = FlatTable(m.nwords, nv.KIND_BACKOFF, start_units=["\x00"], end_chr=m.end_chr, flags=nv.FLAG_NO_DENSE)
    assert len(flat.array(nv.ARR_DENSE)) == 0
    dev = DeviceModel(flat)
    test = list(dict.fromkeys(synth.zipf_corpus(20000, oov_frac=0.03, stream=2))) + ["", "\x7f"]
    syms, off = flat.encode_chars(test)
    a = dev.score_ids(syms, off)
    syms2, off2 = m._flat.encode_chars(test)
    b = m._dev.score_ids(syms2, off2)
    assert torch.equal(a.view(torch.int64), b.view(torch.int64))
    assert np.array_equal(a.cpu().numpy().view(np.uint64), cm.score(test).view(np.uint64))


@pytest.mark.parametrize("index", ["runs", "sorted"])
def test_rank_pipeline_full_size(big_model, index):
    """config-2 sizes: 1e7 samples, 1e6 queries -- size-independent properties plus an
    independent device sort (torch.sort) and a CPU prefix sum on a stride; with the run-length index (the
    default: the sorted keys and positions are written from the runs when looked at) and with the expanded arrays."""
    m, cm = big_model
    n = 10_000_000
    keys = m._dev.sample(n, 99)["ml2p"]
    m._dev.check()
    rk = DeviceRank(keys, index=index)
    assert (rk.runs is not None) == (index == "runs") and rk.check()["error"] == 0
    srt, _ = torch.sort(keys)
    assert torch.equal(rk.keys_sorted.view(torch.int64), srt.view(torch.int64))
    pos = rk.positions
    assert bool((pos[1:] >= pos[:-1]).all())
    w = torch.exp2(srt - np.log2(n))
    ref = torch.cumsum(w, 0)
    assert float(((pos - ref).abs() / ref).max()) < REL
    total = float(w.sum())
    assert abs(float(pos[-1]) - total) <= REL * total
    # queries: sorted test keys, ties, below-min, above-max
    x = torch.cat([srt[::10][:999000], torch.tensor([0.0, 1e-3, 2000.0, float(srt[0]), float(srt[-1])],
                                                    dtype=torch.float64, device=srt.device),
                   srt[:995] + 1e-7])
    qp, qc = rk.query(x.contiguous())
    cnt = torch.searchsorted(srt, x, right=True)
    assert torch.equal(qc, cnt)
    want = torch.where(cnt > 0, pos[(cnt - 1).clamp(min=0)], torch.zeros_like(x))
    if index == "sorted":
        assert torch.equal(qp.view(torch.int64), want.view(torch.int64))
    else:                                                   # (the runs' own sums; the arrays are written from them)
        assert bool(((qp - want).abs() <= REL * want).all())
    # against the C oracle's sequential prefix sum on a subsample of 2e5 keys
    sub = keys[:200000].contiguous()
    rs = DeviceRank(sub, index=index)
    osrt, opos = c_oracle.rank_build(sub.cpu().numpy())
    assert np.array_equal(rs.keys_sorted.cpu().numpy().view(np.uint64), osrt.view(np.uint64))
    assert np.all(np.abs(rs.positions.cpu().numpy() - opos) <= REL * opos)


def test_config3_shard_size():
    """config 3's per-GPU share: 1e10 samples over 8 GPUs = 1.25e9 keys (10 GB per array) from the v2
    6-gram model, through sample -> rank build -> query.  Size-independent properties: the sorted keys
    are a permutation of the samples (wrap-around checksum, equal extremes) in ascending order; the
    positions are monotone and end at the independently reduced sum of weights; a query's count and
    position agree with torch.searchsorted on the sorted keys; chunked sampling == one launch."""
    free, _ = torch.cuda.mem_get_info()
    n = 1_250_000_000
    if free < 60 * 2 ** 30:
        pytest.skip("needs 60 GB of free HBM")
    m = NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6)
    n_global = 8 * n
    keys = m._dev.sample(n, 5, first_id=3 * n)["ml2p"]
    cnt = m._dev.check()
    assert cnt["failed"] == 0 and cnt["draws"] > n
    part = m._dev.sample(1000, 5, first_id=4 * n - 1000)["ml2p"]      # ids beyond 2^32: same values
    assert torch.equal(part.view(torch.int64), keys[-1000:].view(torch.int64))
    rk = DeviceRank(keys, n_global=n_global)
    srt, pos = rk.keys_sorted, rk.positions
    assert int(srt.view(torch.int64).sum()) == int(keys.view(torch.int64).sum())
    assert float(srt[0]) == float(keys.min()) and float(srt[-1]) == float(keys.max())
    step = 250_000_000
    for lo in range(0, n, step):                                      # (bounded temporaries)
        hi = min(n, lo + step + 1)
        assert bool((srt[lo + 1:hi] >= srt[lo:hi - 1]).all())
        assert bool((pos[lo + 1:hi] >= pos[lo:hi - 1]).all())
    total = sum(float(torch.exp2(keys[lo:lo + step] - np.log2(n_global)).sum()) for lo in range(0, n, step))
    assert abs(float(pos[-1]) - total) <= REL * total
    x = torch.cat([srt[::1250][:999000], torch.tensor([0.0, 3000.0], dtype=torch.float64, device=srt.device)])
    qp, qc = rk.query(x.contiguous())
    want_c = torch.searchsorted(srt, x, right=True)
    assert torch.equal(qc, want_c)
    want_p = torch.where(want_c > 0, pos[(want_c - 1).clamp(min=0)], torch.zeros_like(x))
    assert torch.equal(qp.view(torch.int64), want_p.view(torch.int64))


def test_rank_edge_sizes():
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(3)
    for n in (1, 2, 63, 64, 65, 4095, 4096, 4097, 6143, 6144, 6145, 12289, 100001):
        keys = (torch.rand(n, generator=g, dtype=torch.float64) * 60).to(dev)
        if n > 10:
            keys[n // 2:] = keys[: n - n // 2].clone()          # duplicates
        rk = DeviceRank(keys)
        srt, _ = torch.sort(keys)
        assert torch.equal(rk.keys_sorted.view(torch.int64), srt.view(torch.int64)), n
        ref = torch.cumsum(torch.exp2(srt - np.log2(n)), 0)
        assert float(((rk.positions - ref).abs() / ref).max()) < REL, n
    same = torch.full((5000,), 12.5, dtype=torch.float64, device=dev)      # every radix pass trivial
    rk = DeviceRank(same)
    assert torch.equal(rk.keys_sorted, same)
    # a shard weights with the GLOBAL n (multi-GPU decomposition)
    keys = (torch.rand(5000, generator=g, dtype=torch.float64) * 40).to(dev)
    full = DeviceRank(keys)
    a, b = DeviceRank(keys[:2000].contiguous(), n_global=5000), DeviceRank(keys[2000:].contiguous(), n_global=5000)
    x = torch.linspace(0, 45, 777, dtype=torch.float64, device=dev)
    pf, cf = full.query(x)
    pa, ca = a.query(x)
    pb, cb = b.query(x)
    assert torch.equal(cf, ca + cb)
    assert float(((pa + pb - pf).abs() / pf.clamp(min=1e-300)).max()) < REL


def test_rank_key_distributions():
    """Key sets the sampler never produces but the C ABI accepts: the radix sort must not care.
    Many tiles each; checked against torch.sort bit for bit."""
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(11)
    n = 1_000_003
    up = torch.arange(n, dtype=torch.float64) * 1e-4 + 3.0
    cases = {
        "ascending": up,
        "descending": up.flip(0),
        "two values": torch.where(torch.rand(n, generator=g) < 0.999, 17.25, 17.5).double(),
        # one value holds almost every key
        "one heavy bin": torch.where(torch.rand(n, generator=g) < 0.98, torch.tensor(40.0, dtype=torch.float64),
                                     torch.rand(n, generator=g, dtype=torch.float64) * 200),
        # every byte of the key varies: any positive finite double, denormals included
        "all bit patterns": torch.randint(0, 0x7FEFFFFFFFFFFFFF, (n,), generator=g, dtype=torch.int64)
        .view(torch.float64),
        "runs": torch.repeat_interleave(torch.rand(n // 1000 + 1, generator=g, dtype=torch.float64) * 90, 1000)[:n],
        # dense groups far apart, at three scales at once: no single bucket width separates their keys
        "nested clusters": ((torch.randint(0, 5, (n,), generator=g) << 50) + (torch.randint(0, 7, (n,), generator=g) << 25)
                            + torch.randint(0, 1 << 12, (n,), generator=g) + (1 << 60)).view(torch.float64),
        # the words mc_order_keys sorts: (rank of a score with many ties) << 32 | index
        "order words": ((torch.randint(0, 300, (n,), generator=g) ** 2 << 32) + torch.arange(n) + (1 << 52))
        .view(torch.float64),
        # keys whose gaps grow geometrically
        "geometric": ((torch.ones(n, dtype=torch.int64) << torch.randint(8, 61, (n,), generator=g))
                      + torch.randint(0, 200, (n,), generator=g)).view(torch.float64),
    }
    for name, keys in cases.items():
        keys = keys.contiguous().to(dev)
        rk = DeviceRank(keys)
        srt, _ = torch.sort(keys)
        assert torch.equal(rk.keys_sorted.view(torch.int64), srt.view(torch.int64)), name
        if name in ("nested clusters", "order words", "geometric"):
            continue                                         # (bit patterns, not scores: no weights to sum)
        if name != "all bit patterns":                       # (there the weights overflow: 2 ** 1e308)
            ref = torch.cumsum(torch.exp2(srt - np.log2(n)), 0)
            assert float(((rk.positions - ref).abs() / ref).max()) < REL, name


def test_rank_build_takes_the_last_builds_heavy_values_again(big_model):
    """The workspace hands the list of heavy values from one build to the next (include/mi355mc.h, mc_rank_build):
    builds of fresh samples from one model reuse it (age 1, 2, ...) and are searched for again after 63 builds; keys
    from another source in the same workspace are sorted correctly with the stale list and bring a new search at
    the build after; a workspace full of noise is no list.  Sorted keys equal torch.sort every time, positions are
    those of a build that searched (bit for bit: a fresh workspace)."""
    m, _ = big_model
    n = 400_003
    ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    ws.random_(0, 256)                                       # noise: must not pass for a list
    g = torch.Generator(device="cuda").manual_seed(1)

    def build(keys, workspace):
        rk = DeviceRank(keys, workspace=workspace)
        info = rk.check()
        assert torch.equal(rk.keys_sorted, torch.sort(keys)[0])
        return rk, info

    ages = []
    for it in range(67):
        keys = m._dev.sample(n, 100 + it)["ml2p"]
        rk, info = build(keys, ws)
        ages.append(info["list_age"])
        if it in (0, 1, 5, 64):
            fresh = torch.empty_like(ws).zero_()
            rk2, info2 = build(keys, fresh)
            assert info2["list_age"] == 0
            assert torch.equal(rk.positions.view(torch.int64), rk2.positions.view(torch.int64))
    assert ages[:4] == [0, 1, 2, 3] and ages[63] == 63 and ages[64] == 0 and ages[65] == 1, ages
    # another source: uniform keys have no heavy values at all
    other = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 40
    _, a = build(other, ws)
    _, b = build(other, ws)
    assert a["list_age"] == 3 and a["tail_keys"] == n and b["list_age"] == 0, (a, b)
    # and back: the search runs (the uniform keys left no list), then serves again
    keys = m._dev.sample(n, 7)["ml2p"]
    _, c = build(keys, ws)
    _, d = build(m._dev.sample(n, 8)["ml2p"], ws)
    assert c["list_age"] == 0 and d["list_age"] == 1 and d["heavy_values"] == c["heavy_values"], (c, d)


def test_rank_large_search_for_heavy_values(big_model):
    """From 2^26 keys on the rank build counts a 2^19-key sample over several workgroups and takes up to 3072
    values out of the sort (csrc/mc_sort.hip: k_sniff_count .. k_sniff_sort, k_strip<BIG>, k_finish<BIG>,
    k_fill<BIG>); mc_rank_debug_search(1) asks for that from 2^22 keys on, (2) never.  Either way: torch.sort's
    keys bit for bit, and the same positions bit for bit (the scan sees the same sorted keys).  Once at a size
    that takes the large search by itself."""
    from montecarlo_amd import _native as nv
    m, _ = big_model
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    n = (1 << 22) + 4097
    few = torch.randint(0, 2500, (n,), generator=g).double() / 8 + 2
    cases = {
        "model samples": m._dev.sample(n, 7)["ml2p"],
        "all distinct": torch.rand(n, generator=g, dtype=torch.float64) * 90,
        "2500 values": few,
        "one value": torch.full((n,), 7.25, dtype=torch.float64),
        # more candidates than the list holds, every second key distinct
        "10000 values and noise": torch.where(torch.rand(n, generator=g) < 0.5,
                                              torch.randint(0, 10000, (n,), generator=g).double() + 3,
                                              torch.rand(n, generator=g, dtype=torch.float64) * 10000 + 3),
        # heavy values that differ in their lowest bits only, and the empty marker's neighbours
        "adjacent bit patterns": (torch.randint(0, 3000, (n,), generator=g) + 0x4030000000000000).view(torch.float64),
        "top of the range": (0x7FEFFFFFFFFFFFFF - torch.randint(0, 50, (n,), generator=g)).view(torch.float64),
    }
    found = {}
    for name, keys in cases.items():
        keys = keys.contiguous().to(dev)
        want = torch.sort(keys.view(torch.int64))[0]
        got = {}
        for mode in (1, 2):
            nv.lib.mc_rank_debug_search(mode)
            try:
                rk = DeviceRank(keys, index="sorted")
                info = rk.check()
            finally:
                nv.lib.mc_rank_debug_search(0)
            assert info["error"] == 0, (name, mode)
            assert torch.equal(rk.keys_sorted.view(torch.int64), want), (name, mode)
            got[mode] = (rk.positions.view(torch.int64).clone(), info)
        if name != "top of the range":                                      # (weights overflow there: NaN != NaN)
            assert torch.equal(got[1][0], got[2][0]), name
        found[name] = (got[1][1]["heavy_values"], got[2][1]["heavy_values"])
    assert found["model samples"][0] > 1024 >= found["model samples"][1] > 100     # the larger list is in use
    assert found["2500 values"][0] > 2000 and found["all distinct"] == (0, 0) and found["one value"] == (1, 1)
    n = (1 << 26) + 77                                                      # by size
    keys = m._dev.sample(n, 9)["ml2p"]
    rk = DeviceRank(keys, index="sorted")
    info = rk.check()
    assert info["error"] == 0 and info["heavy_values"] > 1024
    assert torch.equal(rk.keys_sorted.view(torch.int64), torch.sort(keys.view(torch.int64))[0])
    # the runs of the heavy values are written 16 bytes a lane: an output array that is only 8-byte aligned, both sizes
    for n, mode in (((1 << 22) + 4097, 1), (1_000_003, 0)):
        keys = m._dev.sample(n, 11)["ml2p"]
        buf = torch.full((n + 3,), -1.0, dtype=torch.float64, device=dev)
        nv.lib.mc_rank_debug_search(mode)
        try:
            rk = DeviceRank(keys, index="sorted", keys_sorted=buf[1:n + 1])
            info = rk.check()
        finally:
            nv.lib.mc_rank_debug_search(0)
        assert info["error"] == 0 and info["heavy_values"] > 100
        assert torch.equal(buf[1:n + 1].view(torch.int64), torch.sort(keys.view(torch.int64))[0])
        assert buf[0].item() == -1.0 and bool((buf[n + 1:] == -1.0).all())


def test_sort_and_order_fuzz():
    """Differential fuzz (scripts/sort_fuzz.py: random sizes up to 3e6, random mixtures of ties, clusters at several
    scales, single-bit keys, full bit patterns, runs, order words) of the key sort and of mc_order_keys against
    torch.sort; 80 seeded cases here, thousands when the script is run by hand."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sort_fuzz", os.path.join(root, "scripts", "sort_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from montecarlo_amd.engine import device_order
    g = torch.Generator().manual_seed(2024)
    for c in range(80):
        n = int(10 ** (torch.rand(1, generator=g).item() * 6.5)) + 1
        x, kind = fz.make(g, n)
        xd = x.cuda()
        rk = DeviceRank(xd, index="sorted")
        assert rk.check()["error"] == 0
        assert torch.equal(rk.keys_sorted.view(torch.int64), torch.sort(xd.view(torch.int64))[0]), (c, kind, n)
        srt, order = device_order(xd)
        want = torch.sort(xd, stable=True)
        assert torch.equal(order.long() & 0xFFFFFFFF, want[1]), (c, kind, n)
        assert torch.equal(srt.view(torch.int64), want[0].view(torch.int64)), (c, kind, n)


def test_distribution_vs_reference_sampler(train_char):
    """(d): the reference's own Mersenne-Twister samples (golden/dist_backwords_thr10.npz) against
    Philox samples from the GPU: two-sample KS on -log2 p, chi-square on string length."""
    from scipy import stats
    ref = np.load(golden_path("dist_backwords_thr10.npz"))
    m = BackWordsMonteCarlo(synth.as_file(train_char), splitter="", threshold=10)
    res = m._dev.sample(200000, 2024, want_len=True)
    m._dev.check()
    ks = stats.ks_2samp(res["ml2p"].cpu().numpy(), ref["ml2p"])
    assert ks.pvalue > 1e-3, ks
    lens = res["len"].cpu().numpy()
    top = int(max(lens.max(), ref["len"].max())) + 1
    a, b = np.bincount(lens, minlength=top).astype(float), np.bincount(ref["len"], minlength=top).astype(float)
    keep = (a + b) >= 50
    table = np.vstack([a[keep], b[keep]])
    chi = stats.chi2_contingency(table)
    assert chi[1] > 1e-3, chi[:2]


@pytest.mark.parametrize("case", ["nwords_n3_char", "nwords_n4_char_min6", "nwords_n2_tab", "backwords_thr10_char",
                                  "backwords_thr3_char_min7", "backwords_thr3_tab"])
def test_enumerators_match_reference(case, train_char, train_tab):
    """nwords_enumerator.enumerator / backwords_enumerator.enumerator on the GPU write, byte for byte, what
    the reference's recursive enumerators wrote (order, strings incl. the appended early end symbols,
    5/8-decimal formatting) and leave -log2 p in the caller's dict as they do."""
    import hashlib
    import math
    from montecarlo_amd.backwords.backwords_enumerator import enumerator as back_enum
    from montecarlo_amd.backwords.backwords_trainer import backwords_counter
    from montecarlo_amd.nwords.nwords_enumerator import enumerator as nw_enum
    from montecarlo_amd.nwords.nwords_trainer import nwords_counter
    fx = golden("enumerators.json")[case]
    lines = train_char if fx["train"] == "train_char.txt" else train_tab
    buf = io.StringIO()
    if fx["kind"] == "nwords":
        probs, _ = nwords_counter(synth.as_file(lines), fx["n"], fx["splitter"], fx["end_chr"], 0, 1,
                                  start_chr=fx["start_chr"])
        first = next(iter(next(iter(probs.values())).values()))
        nw_enum(probs, threshold=fx["threshold"], start_chr=fx["start_chr"], end_chr=fx["end_chr"],
                min_len=fx["min_len"], f_save=buf, order=fx["n"] - 1)
    else:
        probs, _ = backwords_counter(synth.as_file(lines), fx["splitter"], fx["start_chr"], fx["end_chr"], 0, 1,
                                     fx["count_threshold"], 256)
        first = next(iter(next(iter(probs.values())).values()))
        back_enum(probs, fx["threshold"], fx["start_chr"], fx["end_chr"], fx["min_len"], buf)
    assert next(iter(next(iter(probs.values())).values())) == -math.log2(first)
    text = buf.getvalue()
    got = text.split("\n")[:-1]
    assert len(got) == fx["n_lines"]
    if "lines" in fx:
        assert got == fx["lines"]
    assert hashlib.sha256(text.encode("utf-8", "surrogatepass")).hexdigest() == fx["sha256"]


def test_enumeration_at_scale_matches_oracle(big_model):
    """The 100k-line backoff model enumerated to a threshold that yields ~1e5..1e6 strings: the same strings
    and sums, in the same order, as the CPU restatement of the reference's recursion; the node limit is
    an error, not a truncation."""
    from montecarlo_amd import _native as nv
    from oracle import mc_oracle as orc
    m, _ = big_model
    probs = {ctx: {k: vals[k] for k in keys} for ctx, (vals, keys, cum) in m.nwords.items()}
    thr = 24.0
    want = orc.enumerate_strings(probs, thr, "\x00", "\x03", 4)
    ml2p, lens, syms = m._dev.enumerate(thr, 4, 256)
    assert len(ml2p) == len(want) > 50000
    got_s = m._flat.decode(syms, lens)
    assert got_s == [w[0] for w in want]
    assert np.array_equal(ml2p.view(np.uint64), np.array([w[1] for w in want]).view(np.uint64))
    assert bool((ml2p < thr).all())
    with pytest.raises(nv.NativeError, match="max_nodes"):
        m._dev.enumerate(thr, 4, 256, max_nodes=1000)
    # nothing is below the threshold (the reference's iterate returns at once, :27): no strings, no error
    for empty_thr in (0.0, -1.0, 1e-9):
        e_ml2p, e_lens, e_syms = m._dev.enumerate(empty_thr, 4, 256)
        assert len(e_ml2p) == len(e_lens) == len(e_syms) == 0
    # min_len 0: a string may be empty (the end symbol straight from the start context)
    z_ml2p, z_lens, z_syms = m._dev.enumerate(12.0, 0, 256)
    z_want = orc.enumerate_strings(probs, 12.0, "\x00", "\x03", 0)
    assert m._flat.decode(z_syms, z_lens) == [w[0] for w in z_want]
    assert np.array_equal(z_ml2p.view(np.uint64), np.array([w[1] for w in z_want], dtype=np.float64).view(np.uint64))


def golden_path(name):
    import os
    from conftest import GOLDEN
    return os.path.join(GOLDEN, name)


def test_rank_build_failure_path_is_loud_and_stays_inside_its_buffers(monkeypatch):
    """The device-side failure path of mc_rank_build, on demand (mc_rank_debug_fail: the sort's tables
    "overflow" at the first plan): nothing is written outside keys_sorted / positions / workspace (canaries either
    side of each), keys_in is untouched, positions are NaN, mc_rank_check reports it, DeviceRank.check and the
    guess-number entry points raise -- and the same buffers give the right result once the knob is off."""
    import ctypes as C
    from montecarlo_amd import _native as nv
    dev = torch.device("cuda")
    n, pad = 300_000, 4096
    g = torch.Generator(device="cpu").manual_seed(5)
    keys = (torch.rand(n, generator=g, dtype=torch.float64) * 50 + 3).to(dev)
    keys_before = keys.clone()
    need = nv.lib.mc_rank_workspace_bytes(n)
    canary = 0x5A
    srt_buf = torch.full((n + 2 * pad,), -7.0, dtype=torch.float64, device=dev)
    pos_buf = torch.full((n + 2 * pad,), -9.0, dtype=torch.float64, device=dev)
    ws_buf = torch.full((need + 2 * pad * 8,), canary, dtype=torch.uint8, device=dev)
    srt, pos, ws = srt_buf[pad:pad + n], pos_buf[pad:pad + n], ws_buf[pad * 8:pad * 8 + need]
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def build():
        nv.check(nv.lib.mc_rank_build(C.c_void_p(keys.data_ptr()), n, n, C.c_void_p(srt.data_ptr()),
                                      C.c_void_p(pos.data_ptr()), C.c_void_p(ws.data_ptr()), need, stream))

    def canaries_intact():
        return (bool((srt_buf[:pad] == -7.0).all()) and bool((srt_buf[pad + n:] == -7.0).all())
                and bool((pos_buf[:pad] == -9.0).all()) and bool((pos_buf[pad + n:] == -9.0).all())
                and bool((ws_buf[:pad * 8] == canary).all()) and bool((ws_buf[pad * 8 + need:] == canary).all()))

    nv.lib.mc_rank_debug_fail(1)
    try:
        build()
        info = (C.c_uint64 * 8)()
        rc = nv.lib.mc_rank_check(C.c_void_p(ws.data_ptr()), n, stream, info)
        assert rc == -6 and info[0] != 0 and b"failed on the device" in nv.lib.mc_last_error()
        assert bool(torch.isnan(pos).all())
        assert canaries_intact() and torch.equal(keys, keys_before)
        rk = DeviceRank(keys, index="sorted")                      # the host classes surface it
        with pytest.raises(nv.NativeError, match="failed on the device"):
            rk.check()
        monkeypatch.setenv("MC_RANK_INDEX", "sorted")              # (the knob fails mc_rank_build, not the run index)
        with pytest.raises(RuntimeError, match="NaN"):
            MonteCarloLib(keys).ml2p2rank(20.0)
        monkeypatch.delenv("MC_RANK_INDEX")
    finally:
        nv.lib.mc_rank_debug_fail(0)
    build()
    assert nv.lib.mc_rank_check(C.c_void_p(ws.data_ptr()), n, stream, None) == 0
    assert torch.equal(srt.view(torch.int64), torch.sort(keys)[0].view(torch.int64))
    assert bool(torch.isfinite(pos).all()) and canaries_intact()
