#!/usr/bin/env python3
"""The rank build with the small and with the large search for heavy values (mc_rank_debug_search), side by side.
  rank_search_compare.py check            both against torch.sort and against each other on seven key sets at three sizes
  rank_search_compare.py time N [N ...]   both timed on the samples of the v2 and the backoff model at N keys
(profiles/r02_h_rank_build_small_vs_large_search.txt is `time 1e7 67108864 1.25e9`.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import torch
from montecarlo_amd import synth, _native as nv
from montecarlo_amd.engine import DeviceRank
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo

what = sys.argv[1] if len(sys.argv) > 1 else "check"
models = {"v2": lambda: NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6),
          "backoff": lambda: BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)}
built = {k: f() for k, f in models.items()}

def build(keys, mode, reps=0):
    n = keys.numel()
    nv.lib.mc_rank_debug_search(mode)
    try:
        ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device="cuda")
        ks, ps = torch.empty_like(keys), torch.empty_like(keys)
        rk = DeviceRank(keys, n_global=n, workspace=ws, keys_sorted=ks, positions=ps)
        info = rk.check()
        ms = None
        if reps:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                DeviceRank(keys, n_global=n, workspace=ws, keys_sorted=ks, positions=ps)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
        return ks, ps, info, ms
    finally:
        nv.lib.mc_rank_debug_search(0)

if what == "check":
    g = torch.Generator(device="cuda").manual_seed(3)
    for n in (1 << 22, 10_000_019, (1 << 26) + 12345):
        cases = {name: m._dev.sample(n, 7)["ml2p"] for name, m in built.items()}
        cases["uniform"] = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 90
        cases["ten values"] = torch.randint(0, 10, (n,), generator=g, device="cuda").double() * 1.5 + 2
        cases["one value"] = torch.full((n,), 7.25, device="cuda", dtype=torch.float64)
        cases["5000 values zipf"] = (torch.rand(n, generator=g, device="cuda", dtype=torch.float64) ** 6 * 5000).floor() / 7 + 1
        cases["half heavy half distinct"] = torch.where(torch.rand(n, generator=g, device="cuda") < 0.5,
                                                        (torch.randint(0, 3000, (n,), generator=g, device="cuda").double() + 3),
                                                        torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 3000 + 3)
        for name, keys in cases.items():
            want = torch.sort(keys).values
            ks1, ps1, info1, _ = build(keys, 1)
            ks2, ps2, info2, _ = build(keys, 2)
            ok = torch.equal(ks1.view(torch.int64), want.view(torch.int64)) and torch.equal(ks2.view(torch.int64), want.view(torch.int64))
            okp = torch.equal(ps1.view(torch.int64), ps2.view(torch.int64))
            print(f"n={n} {name}: sorted {'OK' if ok else 'WRONG'} positions {'same' if okp else 'DIFFER'} "
                  f"large: heavy={info1['heavy_values']} tail={info1['tail_keys']} err={info1['error']} | small: heavy={info2['heavy_values']} tail={info2['tail_keys']}", flush=True)
            assert ok and okp and info1["error"] == 0
            del want, ks1, ps1, ks2, ps2
        del cases
        torch.cuda.empty_cache()
else:
    for n in [int(float(x)) for x in sys.argv[2:]]:
        for name, m in built.items():
            keys = m._dev.sample(n, 5)["ml2p"]
            out = []
            for mode in (2, 1):
                _, _, info, ms = build(keys, mode, reps=3)
                out.append(f"{'small' if mode == 2 else 'large'} search {ms:.3f} ms ({40*n/ms/1e9:.2f} TB/s; heavy {info['heavy_values']}, tail {info['tail_keys']/n:.3f}, levels {info['levels']})")
                torch.cuda.empty_cache()
            print(f"{name} n={n}: " + " | ".join(out), flush=True)
            del keys
            torch.cuda.empty_cache()
