"""The command lines (montecarlo_amd/_cli.py) against the reference's repo-root scripts: the same flags,
destinations and defaults (read from nwords_simulator.py:119-138, backwords_simulator.py:52-76,
backwords_secondary_simulator.py:28-40, bpe_simulator.py:175-185, nwords_simulator_v2.py:98-105,
nwords_enum.py:7-27, backwords_enum.py:7-17, backwords_secondary_trainer.py:8-30), and -- on the GPU --
the same files as the five calls of the driver flow made by hand."""
import hashlib
import io
import os
import pickle
import random

import pytest

from conftest import GOLDEN
from montecarlo_amd import _cli


REAL_PARSE = _cli.argparse.ArgumentParser.parse_args


def options_of(monkeypatch, front_end, argv):
    """Parse ``argv`` with a front end's own parser and stop before any work is done."""
    seen = {}

    class Stop(Exception):
        pass

    def parse_only(self, args=None, namespace=None):
        seen["args"] = REAL_PARSE(self, args, namespace)
        seen["flags"] = sorted(s for a in self._actions for s in a.option_strings if s not in ("-h", "--help"))
        raise Stop

    monkeypatch.setattr(_cli.argparse.ArgumentParser, "parse_args", parse_only)
    with pytest.raises(Stop):
        front_end(argv)
    got = vars(seen["args"])
    for v in got.values():
        if hasattr(v, "close") and v not in (_cli.sys.stdout,):
            v.close()
    return got, seen["flags"]


TRAIN = os.path.join(GOLDEN, "train_char.txt")
TEST = os.path.join(GOLDEN, "test_char.txt")


def test_flags_and_defaults_are_the_references(monkeypatch, tmp_path):
    out = str(tmp_path / "o.txt")
    got, flags = options_of(monkeypatch, _cli.nwords_simulator, ["-i", TRAIN, "-t", TEST, "-s", out])
    assert flags == sorted(["-i", "--input", "-t", "--test", "-s", "--save", "-n", "--ngram", "--size", "--splitter",
                            "--start4word", "--skip4word"])
    assert (got["ngram"], got["size"], got["splitter"], got["start4word"], got["skip4word"]) == (2, 100000, "\t", 0, 1)
    got, _ = options_of(monkeypatch, _cli.nwords_simulator,
                        ["-i", TRAIN, "-t", TEST, "-s", out, "--splitter", "a\\\\b", "-n", "4"])
    assert got["splitter"] == "a\\b" and got["ngram"] == 4
    with pytest.raises(SystemExit):
        options_of(monkeypatch, _cli.nwords_simulator, ["-i", TRAIN, "-t", TEST, "-s", out, "-n", "7"])

    got, flags = options_of(monkeypatch, _cli.backwords_simulator, ["-i", TRAIN, "-t", TEST, "-s", out])
    assert flags == sorted(["-i", "--input", "-t", "--test", "-s", "--save", "--size", "--splitter", "--start4word",
                            "--skip4word", "--threshold", "--debug-mode", "--max-gram", "--max-iter"])
    assert (got["size"], got["splitter"], got["start4word"], got["skip4word"], got["threshold"], got["debug_mode"],
            got["max_gram"], got["max_iter"]) == (100000, "empty", 0, 1, 10, False, 256, 10 ** 20)

    model = tmp_path / "m.pickle"
    model.write_bytes(b"")
    got, flags = options_of(monkeypatch, _cli.backwords_secondary_simulator, ["-m", str(model), "-t", TEST, "-s", out])
    assert flags == sorted(["-m", "--model", "-t", "--test", "-s", "--save", "--size", "--debug-mode", "--max-iter"])
    assert (got["size"], got["debug_mode"], got["max_iter"]) == (100000, False, 10 ** 20)

    got, flags = options_of(monkeypatch, _cli.bpe_simulator, ["-m", "dir", "-t", TEST, "-s", out])
    assert flags == sorted(["-m", "--model", "-t", "--target", "-s", "--save", "--size"])
    assert (got["model"], got["size"]) == ("dir", 1000000)

    got, flags = options_of(monkeypatch, _cli.nwords_simulator_v2, ["-f", TRAIN, "-t", TEST])
    assert flags == sorted(["-f", "--file", "-t", "--target", "-s", "--save"])
    assert got["save"] is _cli.sys.stdout and got["training"].name == TRAIN

    got, flags = options_of(monkeypatch, _cli.nwords_enum,
                            ["-f", TRAIN, "-n", "3", "--splitter", "empty", "-p", "1e-6", "-s", out])
    assert flags == sorted(["-f", "--pwd-file", "-n", "--ngram", "--splitter", "--start", "--step", "-p", "--min-prob",
                            "-l", "--min-length", "-s", "--save"])
    assert (got["ngram"], got["start4words"], got["skip4words"], got["min_prob"], got["min_len"]) == (3, None, None,
                                                                                                      1e-6, 4)
    with pytest.raises(SystemExit):                  # only the three words are splitters here (nwords_enum.py:13)
        options_of(monkeypatch, _cli.nwords_enum, ["-f", TRAIN, "-n", "3", "--splitter", ",", "-p", "1", "-s", out])

    got, flags = options_of(monkeypatch, _cli.backwords_enum, ["-f", TRAIN, "-p", "1e-6", "-s", out])
    assert flags == sorted(["-f", "--pwd-file", "-p", "--min-prob", "-l", "--min-length", "-s", "--save"])
    assert got["min_len"] == 4

    got, flags = options_of(monkeypatch, _cli.backwords_secondary_trainer, ["-t", TRAIN, "-s", out])
    assert flags == sorted(["-t", "--training", "-s", "--save", "-m", "--model", "--splitter", "--start4word",
                            "--skip4word", "--threshold", "--max-gram"])
    assert (got["model"], got["splitter"], got["start4words"], got["skip4words"], got["threshold"],
            got["max_gram"]) == (None, "empty", 0, 1, 10, 256)
    assert _cli.named_splitter("TAB") == "\t" and _cli.named_splitter("Space") == " " and _cli.named_splitter(",") == ","


def test_modules_carry_the_references_entry_names():
    import montecarlo_amd.backwords_enum as be
    import montecarlo_amd.backwords_secondary_simulator as bss
    import montecarlo_amd.backwords_secondary_trainer as bst
    import montecarlo_amd.backwords_simulator as bs
    import montecarlo_amd.bpe_simulator as bpe
    import montecarlo_amd.nwords_enum as ne
    import montecarlo_amd.nwords_simulator as ns
    import montecarlo_amd.nwords_simulator_v2 as v2
    for mod in (ns, bs, bss, ne, be, bst):
        assert callable(mod.wrapper)
    assert callable(v2.main) and callable(bpe.main) and callable(bpe.wrapper)


def test_secondary_trainer_command_line_trains_and_continues(tmp_path, train_char, capsys):
    """`backwords_secondary_trainer -t A -s m0` then `-t B -m m0 -s m1` (backwords_secondary_trainer.py:31-57):
    the model files hold what the trainer function returns for the same lines, the config carries the marker
    characters, the options and the list of training files, and a model file continues training."""
    import montecarlo_amd.backwords.backwords_secondary_trainer as sec
    from montecarlo_amd.backwords_secondary_trainer import wrapper
    a, b = tmp_path / "a.txt", tmp_path / "b.txt"
    a.write_text("".join(ln + "\n" for ln in train_char[:1200]))
    b.write_text("".join(ln + "\n" for ln in train_char[1200:1700]))
    m0, m1 = str(tmp_path / "m0.pickle"), str(tmp_path / "m1.pickle")
    wrapper(["-t", str(a), "-s", m0, "--threshold", "3", "--max-gram", "5"])
    wrapper(["-t", str(b), "-s", m1, "-m", m0, "--threshold", "3", "--max-gram", "5", "--splitter", "empty"])
    err = capsys.readouterr().err
    assert f"Secondary training based on: {m0}" in err and f"Prior training files: {a}." in err
    sec.IMPORT_PLAIN_COUNTS = False                     # the dict-based restatement of the reference
    try:
        c0, w0 = sec.backwords_counter(train_char[:1200], "", "\x03", "\x00", 0, 1, 5, 3, nwords_dict={}, words={})
        assert type(c0) is dict
        want0 = ({k: dict(v) for k, v in c0.items()}, dict(w0))
        c1, w1 = sec.backwords_counter(train_char[1200:1700], "", "\x03", "\x00", 0, 1, 5, 3, nwords_dict=c0, words=w0)
    finally:
        sec.IMPORT_PLAIN_COUNTS = True
    with open(m0, "rb") as fd:
        got0 = pickle.load(fd)
    with open(m1, "rb") as fd:
        got1 = pickle.load(fd)
    assert type(got0[0]) is dict and (got0[0], got0[1]) == want0
    assert [(k, list(v.items())) for k, v in got1[0].items()] == [(k, list(v.items())) for k, v in c1.items()]
    assert list(got1[1].items()) == list(w1.items())
    assert got0[2] == {"start_chr": "\x03", "end_chr": "\x00", "max_gram": 5, "threshold": 3,
                       "training_list": [str(a)]}
    assert got1[2]["training_list"] == [str(a), str(b)]


def by_hand(model, test_path, seed, **gc_options):
    from montecarlo_amd.lib4mc.MonteCarloLib import MonteCarloLib
    random.seed(seed)
    mc = MonteCarloLib(model.sample(size=3000))
    with open(test_path) as fd:
        mc.ml2p_iter2gc(minus_log_prob_iter=model.parse_file(fd), **gc_options)
    out = io.StringIO()
    mc.write2(out)
    return out.getvalue()


@pytest.mark.gpu
def test_simulator_command_lines_write_the_driver_flows_file(tmp_path, train_char, monkeypatch, capsys):
    from montecarlo_amd.backwords_secondary_simulator import BackWordsSecondaryMonteCarlo
    from montecarlo_amd.backwords_secondary_simulator import wrapper as secondary_cli
    from montecarlo_amd.backwords_secondary_trainer import wrapper as train_cli
    from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
    from montecarlo_amd.backwords_simulator import wrapper as backwords_cli
    from montecarlo_amd.bpe_simulator import BpePcfgSim
    from montecarlo_amd.bpe_simulator import main as bpe_cli
    from montecarlo_amd.nwords_simulator import NWordsMonteCarlo
    from montecarlo_amd.nwords_simulator import wrapper as nwords_cli
    out = tmp_path / "gc.txt"

    random.seed(21)
    nwords_cli(["-i", TRAIN, "-t", TEST, "-s", str(out), "-n", "3", "--size", "3000", "--splitter", "empty"])
    with open(TRAIN) as fd:
        want = by_hand(NWordsMonteCarlo(fd, splitter="", n=3), TEST, 21)
    assert out.read_text() == want and want.count("\n") > 100

    random.seed(22)
    backwords_cli(["-i", TRAIN, "-t", TEST, "-s", str(out), "--size", "3000", "--threshold", "3", "--max-gram", "6"])
    with open(TRAIN) as fd:
        model = BackWordsMonteCarlo(fd, splitter="", threshold=3, max_gram=6, max_iter=10 ** 20)
    assert out.read_text() == by_hand(model, TEST, 22)

    m0 = str(tmp_path / "m0.pickle")
    train_cli(["-t", TRAIN, "-s", m0, "--threshold", "4"])
    random.seed(23)
    secondary_cli(["-m", m0, "-t", TEST, "-s", str(out), "--size", "3000"])
    with open(m0, "rb") as fd:
        model = BackWordsSecondaryMonteCarlo(fd, max_iter=10 ** 20)
    assert out.read_text() == by_hand(model, TEST, 23)

    # --debug-mode: -log2 p of what is typed, `exit` included, then the prompt ends (backwords_simulator.py:82-88)
    typed = iter(["password", "exit"])
    monkeypatch.setattr("builtins.input", lambda prompt: next(typed))
    capsys.readouterr()
    secondary_cli(["-m", m0, "-t", TEST, "-s", str(out), "--debug-mode"])
    printed = capsys.readouterr().out.split()
    assert [float(x) for x in printed] == [model.calc_ml2p("password")[0], model.calc_ml2p("exit")[0]]

    bpe_dir = os.path.join(GOLDEN, "bpe_model")
    random.seed(24)
    bpe_cli(["-m", bpe_dir, "-t", TEST, "-s", str(out), "--size", "3000"])
    assert out.read_text() == by_hand(BpePcfgSim(model_path=bpe_dir), TEST, 24, need_resort=True, add1=True)


@pytest.mark.gpu
def test_enumerator_command_lines_write_what_the_references_wrote(tmp_path):
    """`nwords_enum` / `backwords_enum` with the arguments the golden cases were recorded with from the
    reference's enumerators (tests/golden/make_golden_enum.py): the same file, byte for byte.  `-p` is
    compared with -log2 p, as in the reference (nwords_enumerator.py:27), whatever the flag is called."""
    from conftest import golden
    from montecarlo_amd.backwords_enum import wrapper as backwords_cli
    from montecarlo_amd.nwords_enum import wrapper as nwords_cli
    out = tmp_path / "found.txt"
    for cli, argv, case in (
        (nwords_cli, ["-f", TRAIN, "-n", "3", "--splitter", "empty", "-p", "17", "-s", str(out)], "nwords_n3_char"),
        (nwords_cli, ["-f", TRAIN, "-n", "4", "--splitter", "empty", "--start", "0", "--step", "1", "-p", "14.5",
                      "-l", "6", "-s", str(out)], "nwords_n4_char_min6"),
        (backwords_cli, ["-f", TRAIN, "-p", "16", "-s", str(out)], "backwords_thr10_char"),
    ):
        fx = golden("enumerators.json")[case]
        cli(argv)
        with open(out, newline="") as fd:
            text = fd.read()
        assert text.count("\n") == fx["n_lines"] > 100
        assert hashlib.sha256(text.encode("utf-8", "surrogatepass")).hexdigest() == fx["sha256"]
