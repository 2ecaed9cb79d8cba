"""bench.py's step arrangements at a small size: every one ends in the bench's own assertions on what the host
received (guess numbers strictly increasing, the order a permutation under which the scores ascend, sorted keys,
monotone positions, the sort's error word zero) and prints one JSON line with the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [[], ["--serial"], ["--serial", "--d2h", "inline"], ["--d2h", "overlap"],
                                   ["--d2h-all"], ["--serial", "--sample-wg-per-cu", "7", "--steps", "1", "--warmup", "0"],
                                   ["--rank-index", "sorted"], ["--rank-index", "sorted", "--serial", "--d2h", "inline"],
                                   ["--whole-order"], ["--whole-order", "--d2h-all"], ["--no-fused-query"], ["--order-wait"],
                                   ["--order-wait", "--d2h-all"]],
                         ids=lambda e: " ".join(e) or "default")
def test_bench_step_arrangements(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--samples", "300000", "--tests", "30001", "--train-lines",
           "20000", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["value"] > 0 and line["n_gpus"] == 1 and line["dtype"] == "f64"
    assert line["test_set_on_second_stream"] == ("--serial" not in extra)
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    # the last step's guess numbers as the host received them through THIS arrangement, against the same step computed
    # again on one stream with fresh buffers, and a cut of it against the C oracle (bench.py check_arrangement)
    chk = line["arrangement_check"]
    assert chk["order_and_scores_bit_identical"] and chk["exact_count_equal"]
    assert chk["guess_numbers_max_rel_diff"] <= 1e-9 and chk["of_them_below_2^50"] <= 2
    if "sorted" in extra:
        assert chk["guess_numbers_differing_from_serial_recompute"] == 0
    assert chk["positions_max_rel_diff_vs_oracle"] <= 1e-9 and chk["guess_numbers_differing_from_oracle"] <= 2
    assert line["cold_step_ms"] > 0 and ("--warmup" in extra or line["first_step_ms"] > 0)
    assert (line["stage_ms"]["d2h"] is None) == ("--d2h" not in extra)


def test_two_ranks_on_one_gpu_match_one_rank_over_all_ids(tmp_path):
    """The multi-GPU step before the driver runs it on 8: `bench.py --gpus 2 --rehearse-on-one-gpu` under
    torch.distributed.run (both ranks on cuda:0, the collective over gloo) draws sample ids [2kN, 2kN + N) and
    [2kN + N, 2(k+1)N) in step k, sums the ranks' partial positions and ranks them; one rank drawing
    [2kN, 2(k+1)N) by itself must hand the host the same guess numbers: within 1e-9 (the partial sums add in
    another order), equal where they are below 2^50 except where ceil() sits on an integer."""
    import numpy as np
    common = ["--tests", "30001", "--train-lines", "20000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    two, one = str(tmp_path / "two.npz"), str(tmp_path / "one.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--rehearse-on-one-gpu", "--samples", "150000", "--dump-last", two] + common,
                       capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line2 = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line2["scaling"] == "weak"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--samples", "300000", "--dump-last", one]
                       + common, capture_output=True, text=True, timeout=170, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(two), np.load(one)
    assert np.array_equal(a["order"], b["order"]) and np.array_equal(a["ml2p"].view(np.int64), b["ml2p"].view(np.int64))
    k = int(a["exact"][0])
    assert k == int(b["exact"][0]) and k > 20000
    ga, gb = a["rank"][:k], b["rank"][:k]
    rel = np.abs(ga - gb) / np.maximum(gb, 1)
    assert rel.max() <= 1e-9, rel.max()
    small = gb < (1 << 50)
    assert small.sum() > 1000 and int(np.sum(ga[small] != gb[small])) <= 2
    # (beyond int64 the array carries float64 estimates)
    ea, eb = a["rank"][k:].view(np.float64), b["rank"][k:].view(np.float64)
    assert np.allclose(ea, eb, rtol=1e-9, atol=0)
