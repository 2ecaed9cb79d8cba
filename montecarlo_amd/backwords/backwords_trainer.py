"""Backoff trainer: same contract as the reference's
backwords/backwords_trainer.py:23-78.  Rows of increasing order; a row keeps the
transitions seen at least ``threshold`` times and hands the remaining mass to
its parent row (the context minus its first symbol), already smoothed."""
import sys
from typing import Dict, List, TextIO, Tuple

from .._counting import float_sum, row_total, split_line, tally_lines, transitions


def parse_line(line: str, splitter: str, start4words: int, step4words: int):
    return split_line(line, splitter, start4words, step4words)


def group_by_length(uniq: Dict[Tuple, int]) -> Dict[int, List[Tuple[Tuple, int]]]:
    groups: Dict[int, List[Tuple[Tuple, int]]] = {}
    for secs, c in uniq.items():
        groups.setdefault(len(secs), []).append((secs, c))
    return groups


def backwords_counter(nwords_list: TextIO, splitter: str, start_chr: str, end_chr: str,
                      start4words: int, step4words: int, threshold: int, max_gram: int):
    uniq, words = tally_lines(nwords_list, splitter, start4words, step4words, [start_chr], end_chr)
    zero = {s: c for s, c in words.items() if s != start_chr}
    ztot = row_total(zero)
    table: Dict[Tuple, Dict[str, float]] = {(): {s: c / ztot for s, c in zero.items()}}
    groups = group_by_length(uniq)
    frequent_lengths = [ln for ln, grp in groups.items() if sum(c for _, c in grp) >= threshold]
    top = min(max(frequent_lengths), max(2, max_gram))
    if top == 1:
        print(f"max gram is {top}, fail to model the password dataset", file=sys.stderr)
        sys.exit(-1)
    for n in range(2, top + 1):
        counts = {}
        for ln, grp in groups.items():
            if ln >= n:
                transitions(grp, n - 1, into=counts)
        for ctx, row in counts.items():
            total = row_total(row)
            if total < threshold:
                continue
            kept = {s: c / total for s, c in row.items() if c >= threshold}
            missing = 1 - float_sum(kept.values())
            if missing == 1:
                continue
            if missing > 0:
                for s, p in table[ctx[1:]].items():
                    kept[s] = kept.get(s, 0) + p * missing
            table[ctx] = kept
    return table, words
