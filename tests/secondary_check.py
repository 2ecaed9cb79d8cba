"""Compares the files a `backwords_secondary_main` run saved with the reference's own run
(tests/golden/secondary_main.json, secondary_strategies.json).  Shared by the GPU test and scripts/rehearse_secondary.py."""
import hashlib
import json
import os
import pickle

REL = 1e-9


def check_saved(save: str, fx: dict):
    assert sorted(os.listdir(save)) == sorted(fx["files"])
    # `samples` strategy: a cracked password's guess number is its place among the sampled strings (exact);
    # the other strategies write the Monte Carlo estimate there (prefix sums: 1e-9 relative)
    by_samples = "samples" in fx["meta"]["args"]
    for name, want in fx["files"].items():
        path = os.path.join(save, name)
        if name.endswith(".pickle"):
            with open(path, "rb") as f:
                counts, _words, _cfg = pickle.load(f)                  # written by this test
            rows = [[list(k), sorted(v.items())] for k, v in counts.items()]
            assert hashlib.sha256(json.dumps(rows, ensure_ascii=True).encode()).hexdigest() == want, name
        elif name == "config.json":
            with open(path) as f:
                cfg = json.load(f)
            cfg["training_list"] = cfg["training_list"][1:]
            assert cfg == want
        elif name in ("iter_result.txt", "sectional_result.txt") or (name.startswith("cracked-") and not by_samples):
            # guess numbers come from the prefix sums (1e-9 relative, ceil may flip on an exact integer)
            with open(path, newline="") as f:
                got_lines = f.read().split("\n")
            want_lines = want.split("\n")
            assert len(got_lines) == len(want_lines)
            inexact = 0
            for g, w in zip(got_lines, want_lines):
                if g == w:
                    continue
                gc, wc = g.split("\t"), w.split("\t")
                assert gc[:3] == wc[:3] and gc[4:] == wc[4:], (g, w)
                assert abs(int(gc[3]) - int(wc[3])) <= REL * int(wc[3]) + 1, (g, w)
                inexact += 1
            assert inexact <= 2, name
        else:
            with open(path, newline="") as f:
                assert f.read() == want, name
