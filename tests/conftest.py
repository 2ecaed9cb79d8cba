import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _fresh_rank_choices():
    """DeviceRank remembers, per process, from which size on the run index did not pay (engine.runs_index_wanted):
    every test starts without that memory."""
    try:
        from montecarlo_amd import engine
        engine.forget_rank_choices()
    except Exception:                                   # (no library built: the tests that need it say so)
        pass
    yield


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def golden_lines(name):
    with open(os.path.join(GOLDEN, name), newline="\n") as f:
        return f.read().split("\n")[:-1]


def hx(x):
    return float(x).hex()


def fh(s):
    return float.fromhex(s)


@pytest.fixture(scope="session")
def train_char():
    return golden_lines("train_char.txt")


@pytest.fixture(scope="session")
def train_tab():
    return golden_lines("train_tab.txt")


@pytest.fixture(scope="session")
def train_v2():
    return golden_lines("train_v2.txt")
