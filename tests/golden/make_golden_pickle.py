#!/usr/bin/env python3
"""Writes tests/golden/secondary_round0_reference.pickle: the `(counts, words, config)` model pickle
exactly as the REFERENCE's secondary trainer builds and dumps it (backwords_secondary_trainer.py:23-99,
backwords_secondary_main.py:28-32), from train_char.txt[:2500] at threshold 4 -- round 0 of
secondary_thr4_char.json.  Run in the build container only (imports /root/reference); the pickle holds
builtin dicts, tuples, strings and ints, nothing else."""
import os
import pickle
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

from backwords.backwords_secondary_trainer import backwords_counter  # noqa: E402

with open(os.path.join(HERE, "train_char.txt"), encoding="utf-8") as f:
    lines = [ln.rstrip("\n") for ln in f][:2500]
cfg = {"start_chr": "\x03", "end_chr": "\x00", "max_gram": 256, "threshold": 4, "training_list": []}
counts, words = backwords_counter(list(lines), "", "\x03", "\x00", 0, 1, 256, 4)
assert type(counts) is dict and type(words) is dict
with open(os.path.join(HERE, "secondary_round0_reference.pickle"), "wb") as fd:
    pickle.dump((counts, words, cfg), file=fd)
print(len(counts), "contexts", sum(len(r) for r in counts.values()), "transitions")
