"""Development aid: where the run-length index suits a model's samples and what it costs against mc_rank_build:
sizes 1e6 .. 6.7e7 (one level) on the v2 6-gram, the backoff and the sub-word backoff model."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import _native as nv  # noqa: E402
from montecarlo_amd import synth  # noqa: E402
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo  # noqa: E402
from montecarlo_amd.engine import DeviceRank, RunsBuffers  # noqa: E402
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo  # noqa: E402

dev = torch.device("cuda", 0)
models = {"v2 6-gram": NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6),
          "backoff": BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10),
          "sub-word backoff": BackWordsMonteCarlo(synth.as_file(synth.subword_corpus(100000)), splitter="\t", threshold=10)}
sizes = [int(float(a)) for a in sys.argv[1:]] or [1_000_000, 4_000_000, 10_000_000, 20_000_000, 40_000_000, 67_000_000]


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name, m in models.items():
    for n in sizes:
        keys = torch.empty(n, dtype=torch.float64, device=dev)
        m._dev.sample(n, 5, first_id=3 * n, out=keys)
        srt, pos = torch.empty_like(keys), torch.empty_like(keys)
        ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device=dev)
        ms_sorted = timed(lambda: DeviceRank(keys, n_global=n, workspace=ws, keys_sorted=srt, positions=pos))
        rb = RunsBuffers(n, dev)
        ms_runs = timed(lambda: DeviceRank(keys, n_global=n, runs=rb))
        info = [int(v) for v in rb.info.cpu().tolist()]
        distinct = int((srt[1:] != srt[:-1]).sum()) + 1
        print(f"{name:>17} {n:>10} keys: runs {ms_runs:.3f} ms ({'ok' if info[1] == 0 else 'REFUSED %#x bin %d of %d: %d + %d values' % (info[1], info[6] >> 32, info[6] & 0xFFFFFFFF, info[7] >> 32, info[7] & 0xFFFFFFFF)}), "
              f"expanded {ms_sorted:.3f} ms; {distinct} distinct, heavy {info[3]}, tail {info[4]}", flush=True)
        del keys, srt, pos, ws, rb
        torch.cuda.empty_cache()
