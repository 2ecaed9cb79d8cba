"""The run-length rank index (mc_rank_runs_build / _query / _expand, include/mi355mc.h; csrc/mc_runs.hip): what
MonteCarloLib.__init__ + ml2p2rank (lib4mc/MonteCarloLib.py:10-32) need, one entry per distinct sampled value.

Called through the C ABI and compared with (a) the reference's own recorded run (tests/golden/rank.json: sorted
keys, positions at every run end, probes), (b) the expanded build mc_rank_build + torch.unique_consecutive: values
and cumulative counts exact, cumulative positions within 1e-9 relative (the additions come in another order), on
sampled keys, on synthetic key sets of every shape the sort's tests use, on edge sizes and on the sort's fuzz;
keys that do not suit the index are refused loudly (MC_ERR_UNSUITED, queries answer NaN) and DeviceRank falls
back to the expanded arrays."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import fh, golden, hx
from montecarlo_amd import _native as nv
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
from montecarlo_amd.engine import DeviceRank, RunsBuffers

pytestmark = pytest.mark.gpu
REL = 1e-9
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _vp(t):
    return C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def build_runs(keys, n_global=None, buffers=None):
    """-> (return code of mc_rank_runs_check, host info, buffers)"""
    n = keys.numel()
    r = buffers if buffers is not None else RunsBuffers(n, keys.device)
    nv.check(nv.lib.mc_rank_runs_build(_vp(keys), n, n_global or n, _vp(r.vals), _vp(r.cum_count), _vp(r.cum_pos),
                                       r.capacity, _vp(r.info), _vp(r.ws), r.ws.numel(), _stream(keys.device)))
    h = (C.c_uint64 * 8)()
    rc = nv.lib.mc_rank_runs_check(_vp(r.info), _stream(keys.device), h)
    return rc, list(h), r


def expanded(keys, n_global=None):
    n = keys.numel()
    ws = torch.zeros(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device=keys.device)
    srt, pos = torch.empty_like(keys), torch.empty_like(keys)
    nv.check(nv.lib.mc_rank_build(_vp(keys), n, n_global or n, _vp(srt), _vp(pos), _vp(ws), ws.numel(), _stream(keys.device)))
    torch.cuda.synchronize()
    return srt, pos


def check_against_expanded(name, keys, n_global=None, queries=True, buffers=None):
    """The index of `keys` against the expanded build: values, counts, positions; queries; the arrays written back."""
    n = keys.numel()
    rc, info, r = build_runs(keys, n_global, buffers)
    assert rc == 0, (name, rc, info, nv.lib.mc_last_error())
    D = info[0]
    srt, pos = expanded(keys, n_global)
    bits = srt.view(torch.int64)
    uv, cnt = torch.unique_consecutive(bits, return_counts=True)
    assert D == uv.numel() and info[2] == n, (name, info)
    assert torch.equal(r.vals[:D].view(torch.int64), uv), name
    cc = torch.cumsum(cnt, 0)
    assert torch.equal(r.cum_count[:D], cc), name
    want, got = pos[cc - 1], r.cum_pos[:D]
    assert torch.equal(torch.isnan(got), torch.isnan(want)), name
    fin = torch.isfinite(want)
    if fin.any():
        assert bool(((got[fin] - want[fin]).abs() <= REL * want[fin].abs()).all()), name
        assert bool((got[fin][1:] >= got[fin][:-1]).all()), name
    if not queries:
        return info
    dev = keys.device
    x = torch.cat([keys[: min(n, 5000)], keys[: min(n, 1000)] + 1e-7,
                   torch.tensor([0.0, 1e-3, 1e9], dtype=torch.float64, device=dev)]).contiguous()
    x = torch.sort(x).values
    p1, c1 = torch.empty_like(x), torch.empty(x.numel(), dtype=torch.int64, device=dev)
    p2, c2 = torch.empty_like(x), torch.empty(x.numel(), dtype=torch.int64, device=dev)
    nv.check(nv.lib.mc_rank_query(_vp(srt), _vp(pos), n, _vp(x), x.numel(), _vp(p1), _vp(c1), _stream(dev)))
    nv.check(nv.lib.mc_rank_runs_query(_vp(r.vals), _vp(r.cum_count), _vp(r.cum_pos), _vp(r.info), _vp(x), x.numel(),
                                       _vp(p2), _vp(c2), _stream(dev)))
    torch.cuda.synchronize()
    assert torch.equal(c1, c2), name
    assert torch.equal(c2, torch.searchsorted(srt, x, right=True)), name
    f2 = torch.isfinite(p1)
    assert torch.equal(f2, torch.isfinite(p2)), name
    if f2.any():
        assert bool(((p1[f2] - p2[f2]).abs() <= REL * p1[f2].abs()).all()), name
    ks, ps = torch.empty_like(keys), torch.empty_like(keys)
    nv.check(nv.lib.mc_rank_runs_expand(_vp(r.vals), _vp(r.cum_count), _vp(r.cum_pos), _vp(r.info), n, n_global or n,
                                        _vp(ks), _vp(ps), _stream(dev)))
    torch.cuda.synchronize()
    assert torch.equal(ks.view(torch.int64), bits), name
    fin = torch.isfinite(pos)
    assert torch.equal(fin, torch.isfinite(ps)), name
    if fin.any():
        assert bool(((ps[fin] - pos[fin]).abs() <= REL * pos[fin].abs()).all()), name
    return info


def score_like(g, n, distinct=4000):
    """keys shaped like sampled -log2 p: a few heavy repeated values and a tail of `distinct` values"""
    heavy = torch.tensor([17.25, 21.5, 33.0078125, 40.0], dtype=torch.float64)
    pick = torch.randint(0, 8, (n,), generator=g)
    tail = torch.round(torch.rand(n, generator=g, dtype=torch.float64) * distinct) / 64 + 3
    return torch.where(pick < 4, heavy[pick.clamp(max=3)], tail).contiguous()


@pytest.fixture(scope="module")
def model():
    return BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)


def test_runs_index_matches_reference_run(train_char):
    """The reference's own recorded MonteCarloLib (tests/golden/rank.json, made by importing it): the index's values
    are the distinct sorted keys, its counts the run ends, its positions the reference's positions AT the run ends
    (1e-9), its answers to the probes the reference's `ml2p2rank`."""
    fx = golden("rank.json")
    dev = torch.device("cuda")
    keys = torch.tensor([fh(x) for x in fx["draw_order"]], dtype=torch.float64, device=dev)
    srt = np.array([fh(x) for x in fx["sorted"]])
    pos = np.array([fh(x) for x in fx["positions"]])
    rc, info, r = build_runs(keys)
    assert rc == 0
    D = info[0]
    ends = np.nonzero(np.append(srt[1:] != srt[:-1], True))[0]           # last index of every run
    assert D == len(ends)
    assert [hx(v) for v in r.vals[:D].cpu().numpy()] == [hx(v) for v in srt[ends]]
    assert np.array_equal(r.cum_count[:D].cpu().numpy(), ends + 1)
    got = r.cum_pos[:D].cpu().numpy()
    assert np.all(np.abs(got - pos[ends]) <= REL * pos[ends])
    x = torch.tensor([fh(p[0]) for p in fx["probes"]], dtype=torch.float64, device=dev)
    qp, qc = torch.empty_like(x), torch.empty(x.numel(), dtype=torch.int64, device=dev)
    nv.check(nv.lib.mc_rank_runs_query(_vp(r.vals), _vp(r.cum_count), _vp(r.cum_pos), _vp(r.info), _vp(x), x.numel(),
                                       _vp(qp), _vp(qc), _stream(dev)))
    for (_, want), p, c in zip(fx["probes"], qp.cpu().numpy(), qc.cpu().numpy()):
        est = p if c > 0 else 1
        assert abs(float(est) - fh(want)) <= REL * fh(want)


def test_runs_index_on_sampled_keys(model):
    for n, seed in ((100_000, 3), (1_000_000, 4), (10_000_000, 5)):
        keys = model._dev.sample(n, seed)["ml2p"]
        info = check_against_expanded(f"model[{n}]", keys, n_global=8 * n)
        assert info[3] > 100 and info[4] < n // 2                        # heavy values found, the tail is a fraction
    # DeviceRank takes the index by itself and writes the reference's arrays from it when they are looked at
    rk = DeviceRank(keys)
    assert rk.runs is not None and rk.check()["index"] == "runs"
    assert torch.equal(rk.keys_sorted.view(torch.int64), torch.sort(keys)[0].view(torch.int64))


def test_runs_index_edge_sizes_and_shapes():
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(7)
    for n in (1, 2, 63, 64, 65, 4095, 4096, 4097, 20000, 70001, 300000, 1 << 20, (1 << 21) + 3):
        check_against_expanded(f"scores[{n}]", score_like(g, n).to(dev), n_global=4 * n)
    check_against_expanded("all equal", torch.full((100000,), 12.5, dtype=torch.float64, device=dev))
    k = torch.full((100000,), 12.5, dtype=torch.float64, device=dev)
    k[5], k[99] = float("inf"), 0.0
    check_against_expanded("inf and zero", k)
    # a NaN and the tables' own empty marker among the keys: the index holds them as the last runs, as the sort
    # puts them last (no queries: a NaN key makes the bisection's predicate non-monotone in either form)
    k[77] = float("nan")
    k[78] = torch.tensor([-1], dtype=torch.int64).view(torch.float64).item()
    check_against_expanded("NaN and all ones", k, queries=False)
    # two values; runs in order; many copies of few values in a narrow range
    n = 1_000_003
    check_against_expanded("two values", torch.where(torch.rand(n, generator=g) < 0.999, 17.25, 17.5).double().to(dev))
    runs = torch.repeat_interleave(torch.rand(n // 1000 + 1, generator=g, dtype=torch.float64) * 90, 1000)[:n]
    check_against_expanded("runs", runs.contiguous().to(dev))
    few = (torch.randint(0, 2500, (n,), generator=g).double() / 8 + 2).to(dev)
    check_against_expanded("2500 values", few)
    # values that differ in their lowest bits only: one sub-range of one bin holds them all -- they go through the
    # overflow table, up to its capacity
    adj = (torch.randint(0, 300, (n,), generator=g) + 0x4030000000000000).view(torch.float64).to(dev)
    check_against_expanded("adjacent bit patterns", adj)


def test_runs_index_refuses_keys_that_do_not_repeat(model):
    """Uniform random keys: a bin holds more distinct values than its table.  The build says so (MC_ERR_UNSUITED),
    the index answers NaN, nothing is written outside the buffers (canaries), and DeviceRank falls back to the
    expanded arrays -- then the same buffers serve a model's samples."""
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(9)
    n, pad = 3_000_000, 4096
    keys = (torch.rand(n, generator=g, dtype=torch.float64) * 60 + 5).to(dev)
    cap, need = nv.lib.mc_rank_runs_capacity(n), nv.lib.mc_rank_runs_workspace_bytes(n)
    bufs = {k: torch.full((cap + 2 * pad,), -7.0, dtype=torch.float64, device=dev) for k in ("vals", "cum_pos")}
    bufs["cum_count"] = torch.full((cap + 2 * pad,), -7, dtype=torch.int64, device=dev)
    ws_buf = torch.full((need + 2 * pad,), 0x5A, dtype=torch.uint8, device=dev)
    r = RunsBuffers.__new__(RunsBuffers)
    r.n, r.tdev, r.capacity = n, keys.device, cap
    r.vals, r.cum_pos, r.cum_count = (bufs[k][pad:pad + cap] for k in ("vals", "cum_pos", "cum_count"))
    r.info = torch.zeros(8, dtype=torch.int64, device=dev)
    r.ws = ws_buf[pad:pad + need]
    nv.check(nv.lib.mc_rank_runs_forget(_vp(r.ws), n, _stream(dev)))

    def canaries_intact():
        ok = all(bool((b[:pad] == -7).all()) and bool((b[pad + cap:] == -7).all()) for b in bufs.values())
        return ok and bool((ws_buf[:pad] == 0x5A).all()) and bool((ws_buf[pad + need:] == 0x5A).all())

    rc, info, _ = build_runs(keys, buffers=r)
    assert rc == nv.ERR_UNSUITED and info[0] == 0 and info[1] != 0 and b"do not suit" in nv.lib.mc_last_error()
    assert canaries_intact()
    x = torch.tensor([10.0, 30.0, 70.0], dtype=torch.float64, device=dev)
    qp, qc = torch.zeros_like(x), torch.ones(3, dtype=torch.int64, device=dev)
    nv.check(nv.lib.mc_rank_runs_query(_vp(r.vals), _vp(r.cum_count), _vp(r.cum_pos), _vp(r.info), _vp(x), 3, _vp(qp),
                                       _vp(qc), _stream(dev)))
    assert bool(torch.isnan(qp).all()) and bool((qc == -1).all())
    rk = DeviceRank(keys)
    assert rk.runs is None and rk.check()["index"] == "sorted"
    assert torch.equal(rk.keys_sorted.view(torch.int64), torch.sort(keys)[0].view(torch.int64))
    good = model._dev.sample(n, 11)["ml2p"]
    check_against_expanded("after a refusal", good, buffers=r)
    assert canaries_intact()


def test_runs_workspace_hands_its_hint_on(model):
    """The workspace carries the heavy values and the value range of one build to the next build of n keys: builds
    of fresh samples take them again (age 1, 2, ...), every 64th build searches; another source in the same
    workspace is indexed correctly with the stale hint; a workspace of noise is no hint.  The index never depends on
    it: values and counts equal those of a build that searched, bit for bit; positions within 1e-9 (the bins differ)."""
    n = 400_003
    dev = torch.device("cuda")
    r = RunsBuffers(n, dev)
    r.ws.random_(0, 256)                                     # noise: must not pass for a hint
    ages = []
    for it in range(67):
        keys = model._dev.sample(n, 100 + it)["ml2p"]
        info = check_against_expanded(f"build {it}", keys, buffers=r, queries=it in (0, 1, 64))
        ages.append(info[5])
    assert ages[:4] == [0, 1, 2, 3] and ages[63] == 63 and ages[64] == 0 and ages[65] == 1, ages
    g = torch.Generator(device="cpu").manual_seed(1)
    other = score_like(g, n, distinct=900).to(dev)           # another source: the stale hint still serves
    a = check_against_expanded("another source", other, buffers=r)
    b = check_against_expanded("another source again", other, buffers=r)
    assert a[5] == 3 and b[5] == 0, (a, b)                   # ... and the tail it left brings a new search
    c = check_against_expanded("and back", model._dev.sample(n, 7)["ml2p"], buffers=r)
    assert c[0] > 1000
    # a source far outside the range the hint describes: the build may refuse such keys (they all fall into the
    # last bin) -- loudly, never wrongly -- and the build after that searches again and takes them
    far = model._dev.sample(n, 8)["ml2p"] + 1000.0
    rc, info, _ = build_runs(far, buffers=r)
    assert rc in (0, nv.ERR_UNSUITED), (rc, info)
    d = check_against_expanded("far away, searched for", far, buffers=r)
    assert d[5] == 0 or rc == 0


def test_runs_index_fuzz():
    """The sort's differential fuzz (scripts/sort_fuzz.py: ties of every multiplicity, clusters at several scales,
    single-bit keys, full bit patterns, runs, order words) through DeviceRank as the host classes use it: whichever
    index it settles on, the sorted keys are torch.sort's; where it is the run index, its values and counts are
    torch.unique_consecutive's."""
    spec = importlib.util.spec_from_file_location("sort_fuzz", os.path.join(ROOT, "scripts", "sort_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    g = torch.Generator().manual_seed(77)
    kinds = {"runs": 0, "sorted": 0}
    for c in range(60):
        n = int(10 ** (torch.rand(1, generator=g).item() * 6.3)) + 1
        x, kind = fz.make(g, n)
        xd = x.cuda()
        rk = DeviceRank(xd)
        info = rk.check()
        kinds[info["index"]] += 1
        want = torch.sort(xd.view(torch.int64))[0]
        assert torch.equal(rk.keys_sorted.view(torch.int64), want), (c, kind, n, info)
        if rk.runs is not None:
            uv, cnt = torch.unique_consecutive(want, return_counts=True)
            D = info["distinct_values"]
            assert D == uv.numel() and torch.equal(rk.runs.vals[:D].view(torch.int64), uv), (c, kind, n)
            assert torch.equal(rk.runs.cum_count[:D], torch.cumsum(cnt, 0)), (c, kind, n)
    assert kinds["runs"] >= 15 and kinds["sorted"] >= 5, kinds


def test_runs_index_in_two_levels_beyond_2_26_keys(model):
    """mc_rank_runs_build above mc_rank_runs_max_keys() (2^26): level 1 counts the sample's heavy values and hands every
    other key to one of 64 value-range groups, each group is built like a small input (all side by side), the groups'
    runs and the level-1 heavy values are joined.  7e7 samples of the headline model against the expanded build:
    values, cumulative counts, positions, queries, the arrays written back.  Uniform random keys are refused
    (MC_ERR_UNSUITED, NaN answers, canaries around every buffer intact).  Nothing takes this build by default."""
    dev = torch.device("cuda")
    n = 70_000_000
    assert n > nv.lib.mc_rank_runs_max_keys() and nv.lib.mc_rank_runs_workspace_bytes(n) > 0
    keys = torch.empty(n, dtype=torch.float64, device=dev)
    model._dev.sample(n, 12, first_id=5 * n, out=keys)
    info = check_against_expanded("two levels", keys, n_global=4 * n)
    assert info[1] == 0 and info[3] > 1000 and 0 < info[4] < n // 2      # heavy values of both levels, a tail under half
    assert DeviceRank(keys[: 1 << 20]).runs is not None and DeviceRank(keys).runs is None    # (the default: by size)
    del keys
    torch.cuda.empty_cache()
    g = torch.Generator(device="cpu").manual_seed(10)
    keys = (torch.rand(n, generator=g, dtype=torch.float64) * 60 + 5).to(dev)
    pad = 4096
    cap, need = nv.lib.mc_rank_runs_capacity(n), nv.lib.mc_rank_runs_workspace_bytes(n)
    bufs = {k: torch.full((cap + 2 * pad,), -7.0, dtype=torch.float64, device=dev) for k in ("vals", "cum_pos")}
    bufs["cum_count"] = torch.full((cap + 2 * pad,), -7, dtype=torch.int64, device=dev)
    ws_buf = torch.full((need + 2 * pad,), 0x5A, dtype=torch.uint8, device=dev)
    r = RunsBuffers.__new__(RunsBuffers)
    r.n, r.tdev, r.capacity = n, keys.device, cap
    r.vals, r.cum_pos, r.cum_count = (bufs[k][pad:pad + cap] for k in ("vals", "cum_pos", "cum_count"))
    r.info = torch.zeros(8, dtype=torch.int64, device=dev)
    r.ws = ws_buf[pad:pad + need]
    nv.check(nv.lib.mc_rank_runs_forget(_vp(r.ws), n, _stream(dev)))
    rc, info, _ = build_runs(keys, buffers=r)
    assert rc == nv.ERR_UNSUITED and info[0] == 0 and info[1] != 0 and b"do not suit" in nv.lib.mc_last_error()
    assert all(bool((b[:pad] == -7).all()) and bool((b[pad + cap:] == -7).all()) for b in bufs.values())
    assert bool((ws_buf[:pad] == 0x5A).all()) and bool((ws_buf[pad + need:] == 0x5A).all())
    x = torch.tensor([10.0, 30.0, 70.0], dtype=torch.float64, device=dev)
    qp, qc = torch.zeros_like(x), torch.ones(3, dtype=torch.int64, device=dev)
    nv.check(nv.lib.mc_rank_runs_query(_vp(r.vals), _vp(r.cum_count), _vp(r.cum_pos), _vp(r.info), _vp(x), 3, _vp(qp),
                                       _vp(qc), _stream(dev)))
    assert bool(torch.isnan(qp).all()) and bool((qc == -1).all())


def test_a_refusal_is_remembered_by_size():
    """Samples of a sub-word model (a million distinct values in 1e7 samples): the run index refuses them (here 4e6
    samples), DeviceRank builds the expanded arrays instead and remembers the size (engine.runs_index_wanted): the
    next default build of that many keys goes to mc_rank_build directly, smaller ones still try the index, and
    forget_rank_choices() starts over."""
    from montecarlo_amd import engine
    m = BackWordsMonteCarlo(synth.as_file(synth.subword_corpus(100000)), splitter="\t", threshold=10)
    n = 4_000_000
    keys = m._dev.sample(n, 5, first_id=3 * n)["ml2p"]
    rc, info, _ = build_runs(keys)
    assert rc == nv.ERR_UNSUITED and info[1] & 1                # (a bin of more values than its tables take)
    assert engine.runs_index_wanted(n)
    first = DeviceRank(keys)
    assert first.runs is None and first.check()["index"] == "sorted"
    assert torch.equal(first.keys_sorted.view(torch.int64), torch.sort(keys)[0].view(torch.int64))
    assert not engine.runs_index_wanted(n) and not engine.runs_index_wanted(2 * n) and engine.runs_index_wanted(n - 1)
    engine.forget_rank_choices()
    assert engine.runs_index_wanted(n)


def test_query_inside_the_guess_numbers_equals_the_two_calls(model):
    """mc_rank_runs_guess_numbers (ml2p_iter2gc as the reference writes it: the bisection inside the loop) against
    mc_rank_runs_query followed by mc_guess_numbers: positions, counts, guess numbers and the count of exact ones bit
    for bit -- ascending queries of several sizes (chunk borders, a last partial wave), queries below every key and
    above every key, add1 on and off, a NaN query, and a failed index (NaN positions, no exact guess number)."""
    from montecarlo_amd.engine import device_guess_numbers, device_query_guess_numbers
    dev = torch.device("cuda")
    keys = model._dev.sample(3_000_000, 31)["ml2p"]
    rk = DeviceRank(keys, n_global=3 * keys.numel(), index="runs")
    assert rk.check()["error"] == 0
    g = torch.Generator(device="cpu").manual_seed(4)
    for t in (1, 63, 64, 65, 2047, 2048, 2049, 100_003, 1_000_000):
        x = torch.sort(torch.cat([keys[torch.randint(0, keys.numel(), (t,), generator=g).to(dev)][: max(1, t - 3)],
                                  torch.tensor([0.5, 200.0, 1022.0], dtype=torch.float64, device=dev)])[:t]).values.contiguous()
        for add1 in (True, False):
            p0, c0 = rk.query(x)
            r0, e0, _ = device_guess_numbers(p0, c0, add1)
            p1, c1, r1, e1 = device_query_guess_numbers(rk, x, add1)
            torch.cuda.synchronize()
            assert torch.equal(p0.view(torch.int64), p1.view(torch.int64)) and torch.equal(c0, c1), (t, add1)
            assert torch.equal(r0, r1) and int(e0.item()) == int(e1.item()), (t, add1)
    x = torch.tensor([5.0, float("nan"), 30.0], dtype=torch.float64, device=dev)       # (no order: any x is answered)
    p0, c0 = rk.query(x)
    r0, e0, _ = device_guess_numbers(p0, c0, True)
    p1, c1, r1, e1 = device_query_guess_numbers(rk, x, True)
    assert torch.equal(p0.view(torch.int64), p1.view(torch.int64)) and torch.equal(c0, c1) and torch.equal(r0, r1)
    noise = (torch.rand(3_000_000, dtype=torch.float64) * 60 + 5).to(dev)
    bad = DeviceRank(noise, runs=RunsBuffers(3_000_000, noise.device))
    x = torch.tensor([10.0, 20.0, 30.0], dtype=torch.float64, device=dev)
    p1, c1, r1, e1 = device_query_guess_numbers(bad, x, True)
    assert bool(torch.isnan(p1).all()) and bool((c1 == -1).all()) and int(e1.item()) == 0       # (never the estimate 1)
