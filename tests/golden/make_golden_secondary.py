#!/usr/bin/env python3
"""Golden runs of the reference's secondary-training driver (backwords_secondary_main.py):
`--strategy samples 2` -> tests/golden/secondary_main.json; `guesses`, `hits` and `auto_hits` ->
tests/golden/secondary_strategies.json.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_secondary.py

The reference's `wrapper()` is run as shipped on tests/golden/train_char.txt / test_char.txt.  Its
uniforms are injected as in make_golden.py: `random.uniform` (looked up at call time by
lib4mc/ProbLib.py:41) is replaced by the Philox stream of oracle/philox.py, one stream per sample;
every model instance takes its stream seed from `random.getrandbits(64)` at its first sample, which is
what montecarlo_amd's models do, so `random.seed(SEED)` pins both runs.  Output files are recorded as
text; pickles as a digest of the final counts."""
import hashlib
import json
import os
import random
import sys
import tempfile

os.environ["TQDM_DISABLE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle.philox import SampleStream  # noqa: E402

import backwords_secondary_main as ref_main  # noqa: E402
from backwords_secondary_simulator import BackWordsSecondaryMonteCarlo  # noqa: E402

SEED = 20250304
ARGS = ["--strategy", "samples", "2", "--size", "3000", "--threshold", "4", "--secondary-sample", "150"]


class Feed:
    stream = None

    def uniform(self, a, b):
        return a + (b - a) * self.stream()


FEED = Feed()
random.uniform = FEED.uniform
_plain_sample1 = BackWordsSecondaryMonteCarlo.sample1


def sample1_with_stream(self):
    if not hasattr(self, "_golden_seed"):
        self._golden_seed, self._golden_next = random.getrandbits(64), 0
    FEED.stream = SampleStream(self._golden_seed, self._golden_next)
    self._golden_next += 1
    return _plain_sample1(self)


BackWordsSecondaryMonteCarlo.sample1 = sample1_with_stream


# the other strategies of the driver (reference backwords_secondary_main.py:139-163): guess-number bounds, hit
# bounds, generated hit bounds -> tests/golden/secondary_strategies.json
OTHER = {
    "guesses": ["--strategy", "guesses", "2000", "100000", "--size", "3000", "--threshold", "4",
                "--secondary-sample", "150"],
    "hits": ["--strategy", "hits", "30", "60", "--size", "3000", "--threshold", "4"],
    "auto_hits": ["--strategy", "auto_hits", "2", "15", "100", "--size", "2500", "--threshold", "4",
                  "--secondary-sample", "100"],
}


def run(args):
    import pickle
    out = {"meta": {"seed": SEED, "args": args, "train": "train_char.txt", "test": "test_char.txt"}, "files": {}}
    with tempfile.TemporaryDirectory() as tmp:
        save = os.path.join(tmp, "save")
        sys.argv = ["backwords_secondary_main.py", "-i", os.path.join(HERE, "train_char.txt"),
                    "-t", os.path.join(HERE, "test_char.txt"), "-s", save] + args
        random.seed(SEED)
        ref_main.wrapper()
        for name in sorted(os.listdir(save)):
            path = os.path.join(save, name)
            if name.endswith(".pickle"):
                with open(path, "rb") as f:
                    counts, words, _cfg = pickle.load(f)      # written by this very run
                rows = [[list(k), sorted(v.items())] for k, v in counts.items()]
                out["files"][name] = hashlib.sha256(json.dumps(rows, ensure_ascii=True).encode()).hexdigest()
            elif name != "config.json":
                with open(path, newline="") as f:
                    out["files"][name] = f.read()
            else:
                with open(path) as f:
                    cfg = json.load(f)
                cfg["training_list"] = cfg["training_list"][1:]          # [0] is an absolute path
                out["files"][name] = cfg
    return out


def main():
    out = run(ARGS)
    with open(os.path.join(HERE, "secondary_main.json"), "w") as f:
        json.dump(out, f, ensure_ascii=True, separators=(",", ":"))
    print({k: (len(v) if isinstance(v, str) else v) for k, v in out["files"].items()})
    others = {name: run(args) for name, args in OTHER.items()}
    with open(os.path.join(HERE, "secondary_strategies.json"), "w") as f:
        json.dump(others, f, ensure_ascii=True, separators=(",", ":"))
    for name, o in others.items():
        print(name, {k: (len(v) if isinstance(v, str) else v) for k, v in o["files"].items()})


if __name__ == "__main__":
    main()
