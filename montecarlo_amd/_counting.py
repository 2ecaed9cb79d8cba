"""Host-side n-gram counting shared by the trainer modules.

The trainers' output feeds the table flattener, and its dict ORDER defines the
CDF the sampler inverts, so the counting below keeps the reference's first-seen
orders (nwords/nwords_trainer.py:30-44, backwords/backwords_trainer.py:31-62):
unique lines in first-seen order, positions left to right.  Float arithmetic is
spelled as explicit left-to-right loops: CPython 3.12 changed ``sum()`` to
compensated summation, which would move probabilities by an ulp.
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, List, Tuple

Counts = Dict[Tuple, Dict[str, int]]


def read_lines(source) -> Iterable[str]:
    """Accept a text file (closed afterwards, as the reference trainers do,
    nwords_trainer.py:39) or any iterable of lines."""
    if hasattr(source, "read"):
        try:
            if source.seekable():
                source.seek(0)
            for line in source:
                yield line
        finally:
            source.close()
    else:
        yield from source


def split_line(line: str, splitter: str, first: int, step: int) -> List[str]:
    body = line.strip("\r\n")
    if splitter == "":
        return list(body)
    return re.split(splitter, body)[first::step]


def tally_lines(source, splitter: str, first: int, step: int, lead: List[str], end_chr: str):
    """-> (unique section tuples with multiplicity, first-seen order; symbol frequencies)."""
    uniq: Dict[Tuple[str, ...], int] = {}
    freq: Dict[str, int] = {}
    for line in read_lines(source):
        secs = tuple(lead + split_line(line, splitter, first, step) + [end_chr])
        for s in secs:
            freq[s] = freq.get(s, 0) + 1
        uniq[secs] = uniq.get(secs, 0) + 1
    return uniq, freq


def transitions(uniq_items: Iterable[Tuple[Tuple[str, ...], int]], order: int, into: Counts = None) -> Counts:
    """Count (context of ``order`` symbols -> next symbol) over every position."""
    table: Counts = {} if into is None else into
    for secs, mult in uniq_items:
        for i in range(order, len(secs)):
            row = table.get(secs[i - order:i])
            if row is None:
                row = table[secs[i - order:i]] = {}
            nxt = secs[i]
            row[nxt] = row.get(nxt, 0) + mult
    return table


def row_total(row: Dict[str, int]) -> int:
    t = 0
    for v in row.values():
        t += v
    return t


def float_sum(values) -> float:
    acc = 0
    for v in values:
        acc = acc + v
    return acc
