"""Development aid / evidence: the run-length rank index beyond 2^26 keys (two levels: mc_runs.hip, "builds of more
than 2^26 keys") against the expanded build (mc_rank_build) on the same keys: every query's count equal, positions
within 1e-12, the number of runs = the number of distinct keys; and the time of both builds.
  python scripts/runs_big_check.py [n ...]      (default 7e7 2e8 1.25e9; keys: samples of the synthetic v2 6-gram model)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import _native as nv  # noqa: E402
from montecarlo_amd import synth  # noqa: E402
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo  # noqa: E402
from montecarlo_amd.engine import DeviceRank, RunsBuffers  # noqa: E402
from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo  # noqa: E402

sizes = [int(float(a)) for a in sys.argv[1:]] or [70_000_000, 200_000_000, 1_250_000_000]
dev = torch.device("cuda", 0)
models = {"v2 6-gram": NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6),
          "backoff": BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)}


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for n in sizes:
    for name, m in models.items():
        keys = torch.empty(n, dtype=torch.float64, device=dev)
        m._dev.sample(n, 5, first_id=3 * n, out=keys)
        torch.cuda.synchronize()
        srt, pos = torch.empty_like(keys), torch.empty_like(keys)
        ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n), dtype=torch.uint8, device=dev)
        ms_sorted = timed(lambda: DeviceRank(keys, n_global=8 * n, workspace=ws, keys_sorted=srt, positions=pos))
        ref = DeviceRank(keys, n_global=8 * n, workspace=ws, keys_sorted=srt, positions=pos)
        assert ref.check()["error"] == 0
        rb = RunsBuffers(n, dev)
        ms_runs = timed(lambda: DeviceRank(keys, n_global=8 * n, runs=rb))
        rk = DeviceRank(keys, n_global=8 * n, runs=rb)
        try:
            info = rk.check()
        except nv.NativeError as e:
            words = [int(v) for v in rb.info.cpu().numpy().view(np.uint64)]
            print(f"{n} keys of the {name} model: REFUSED: {e}; info {[hex(v) for v in words]}", flush=True)
            import ctypes as C
            out = (C.c_uint64 * (64 * 9))()
            nv.check(nv.lib.mc_rank_runs_debug_groups(C.c_void_p(rb.ws.data_ptr()), n, out, None))
            for g in range(64):
                r = list(out[g * 9:g * 9 + 9])
                print(f"  group {g:2d}: keys {r[0]:>9} runs {r[1]:>7} err {r[2]:#x} counted {r[3]:>9} heavy {r[4]:>5} tail {r[5]:>9} "
                      f"{r[7]:#x} {r[8]:#x}")
            raise
        # probes: sorted keys at a stride (run ends and insides), midpoints between neighbours, the extremes, beyond
        x = srt[:: max(1, n // 400_000)][:400_000]
        mid = (x[1:] + x[:-1]) * 0.5
        q = torch.cat([x, mid, srt[:1] - 1.0, srt[-1:] + 1.0, srt[:1], srt[-1:]]).contiguous()
        p0, c0 = ref.query(q)
        p1, c1 = rk.query(q)
        torch.cuda.synchronize()
        same_cnt = bool(torch.equal(c0, c1))
        rel = float(((p0 - p1).abs() / p0.abs().clamp_min(1e-300)).max())
        distinct = int((srt[1:] != srt[:-1]).sum()) + 1
        print(f"{n:>11} keys of the {name} model: runs {ms_runs:.3f} ms, expanded arrays {ms_sorted:.3f} ms; "
              f"{info['distinct_values']} runs ({distinct} distinct keys), counts equal: {same_cnt}, positions' largest "
              f"relative difference {rel:.2e}; {info}", flush=True)
        assert info["index"] == "runs" and info["distinct_values"] == distinct and same_cnt and rel <= 1e-12
        del keys, srt, pos, ws, rb, rk, ref, x, mid, q, p0, p1, c0, c1
        torch.cuda.empty_cache()
print("runs_big_check ok")
