"""Fixed-order n-gram trainer: same contract as the reference's
nwords/nwords_trainer.py:22-53 (returns ``({context: {symbol: p}}, words)`` with
the reference's dict order), counting done by montecarlo_amd._counting."""
from typing import Dict, TextIO, Tuple

from .._counting import row_total, split_line, tally_lines, transitions


def parse_line(line: str, splitter: str, start4words: int, skip4words: int):
    return split_line(line, splitter, start4words, skip4words)


def nwords_counter(nwords_list: TextIO, n: int, splitter: str, end_chr: str, start4words: int,
                   skip4words: int, start_chr: str = '\x00') -> Tuple[Dict[Tuple, Dict[str, float]], Dict[str, int]]:
    uniq, words = tally_lines(nwords_list, splitter, start4words, skip4words, [start_chr] * (n - 1), end_chr)
    counts = transitions(uniq.items(), n - 1)
    probs = {}
    for ctx, row in counts.items():
        total = row_total(row)
        probs[ctx] = {sym: c / total for sym, c in row.items()}
    return probs, words
