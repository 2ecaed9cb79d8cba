"""A slice of the differential fuzzers of scripts/ in every GPU test run (the campaigns of thousands of cases are run
by hand: DESIGN.md 4): random corpora through the trainers, the table compiler, k_sample / k_score / k_score_words /
k_sample_bpe / k_score_bpe, parse_file and the enumerators against the oracles (model_fuzz.py); MonteCarloLib ->
ml2p_iter2gc -> write2 against RankOracle (flow_fuzz.py); the run index against the expanded build (runs_fuzz.py).
Each script prints a line per case and exits 0 when every case agreed."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def fresh_rank_choices():
    """DeviceRank remembers the size from which the run index was refused (engine.runs_index_wanted): what the
    fuzzers' key sets taught it must not reach the tests that follow."""
    from montecarlo_amd import engine
    engine.forget_rank_choices()
    yield
    engine.forget_rank_choices()


def run_script(name, *args):
    path = os.path.join(ROOT, "scripts", name)
    spec = importlib.util.spec_from_file_location(name[:-3] + "_under_test", path)
    mod = importlib.util.module_from_spec(spec)
    argv = sys.argv
    sys.argv = [path] + [str(a) for a in args]
    try:
        spec.loader.exec_module(mod)          # (runs_fuzz.py works at module level)
        if hasattr(mod, "main"):
            with pytest.raises(SystemExit) as e:
                mod.main()
            assert e.value.code == 0, f"{name} {args}: a case differs (see the captured output)"
    finally:
        sys.argv = argv


def test_model_side_fuzz(capsys):
    run_script("model_fuzz.py", 40, 101)
    out = capsys.readouterr().out
    assert "ALL OK" in out
    for kind in ("backwords thr", "nwords n=", "sub-word", " v2 n=", "bpe-pcfg"):
        assert kind in out, kind              # every kind of case took part


def test_estimator_flow_fuzz(capsys):
    run_script("flow_fuzz.py", 60, 1)            # (its first case is one the run index refuses: both kinds of index)
    out = capsys.readouterr().out
    assert "ALL OK" in out and "index=runs" in out and "index=sorted" in out and "write2=ok" in out


def test_run_index_fuzz(capsys):
    run_script("runs_fuzz.py", 120, 103)
    assert ": OK" in capsys.readouterr().out
