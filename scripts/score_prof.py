"""Development aid: the two scoring kernels of bench.py's `other_configs` alone, for rocprofv3 passes.
  python scripts/score_prof.py [words|bpe|both] [reps]
k_score_words: 1e6 test passwords through calc_ml2p's DFS on the 100k-line tab-split backoff model;
k_score_bpe: 1e6 test passwords on the synthetic BPE-PCFG model.  Prints device time per launch (HIP events)."""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import _native as nv  # noqa: E402
from montecarlo_amd import synth  # noqa: E402
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo  # noqa: E402
from montecarlo_amd.bpe_simulator import BpePcfgSim  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


if which in ("words", "both"):
    sub = synth.subword_corpus(100000)
    model = BackWordsMonteCarlo(synth.as_file(sub), splitter="\t", threshold=10)
    test = ["".join(ln.split("\t")) for ln in synth.subword_corpus(1_000_000, stream=1)]
    cps, off, max_len = model._flat.encode_text(test)
    d_cps = torch.from_numpy(np.ascontiguousarray(cps, dtype=np.uint32).view(np.int32)).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.int64)).to(dev)
    sc = torch.empty(len(test), dtype=torch.float64, device=dev)
    ws = torch.empty(max(1, nv.lib.mc_score_words_workspace_bytes(model._dev._h, len(test), max_len)), dtype=torch.uint8,
                     device=dev)
    model._dev.check()
    ms = timed(lambda: model._dev.score_words_device(d_cps, d_off, len(test), max_len, 10 ** 20, sc, ws))
    c = model._dev.check()
    print(f"k_score_words: {ms:.4f} ms for {len(test)} passwords (max {max_len} chars), "
          f"{c['score_probes'] / (reps + 1) / len(test):.3f} transition probes and "
          f"{c['score_steps'] / (reps + 1) / len(test):.3f} steps a password; "
          f"sum of scores {float(sc.sum()):.6f}", flush=True)
    if os.environ.get("SCORE_PROF_CHECK"):          # against the library's host-visible wrapper on a cut
        want = np.array([model.calc_ml2p(p) for p in test[:2000]])
        got = sc[:2000].cpu().numpy()
        print("cut of 2000 equal to calc_ml2p:", bool(np.array_equal(want, got)))

if which in ("bpe", "both"):
    with tempfile.TemporaryDirectory() as tmp:
        synth.zipf_bpe_dir(tmp, np.random.default_rng(4))
        m4 = BpePcfgSim(tmp)
    res = m4._dev.sample(200_000, 5, want_len=True, sym_stride=m4._flat.max_segments)
    pw = list(dict.fromkeys(m4._flat.decode_entries(res["syms"].cpu().numpy(), res["len"].cpu().numpy())))
    pw = (pw * (1_000_000 // len(pw) + 1))[:1_000_000]
    cps, cls, off = m4.encode(pw)
    d_cps = torch.from_numpy(np.ascontiguousarray(cps, dtype=np.uint32).view(np.int32)).to(dev)
    d_cls = torch.from_numpy(np.ascontiguousarray(cls, dtype=np.uint8)).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.int64)).to(dev)
    sc = torch.empty(len(pw), dtype=torch.float64, device=dev)
    m4._dev.check()
    ms = timed(lambda: m4._dev.score_bpe_device(d_cps, d_cls, d_off, len(pw), sc))
    c = m4._dev.check()
    print(f"k_score_bpe: {ms:.4f} ms for {len(pw)} passwords, {c['score_probes'] / (reps + 1) / len(pw):.3f} transition "
          f"probes and {c['score_steps'] / (reps + 1) / len(pw):.3f} segments a password; sum of scores "
          f"{float(sc.sum()):.6f}", flush=True)
