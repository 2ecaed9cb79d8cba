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
                                   ["--d2h-all"], ["--serial", "--sample-wg-per-cu", "7", "--steps", "1", "--warmup", "0"]],
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
