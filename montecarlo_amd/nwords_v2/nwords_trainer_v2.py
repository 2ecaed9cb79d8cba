"""n-words v2 trainer: same contract as the reference's
nwords_v2/nwords_trainer_v2.py:12-56.  Contexts are the last n-1 CHARACTERS of
the text so far; transitions are words (words seen fewer than ``threshold``
times are spelled out character by character)."""
from typing import Dict, TextIO

from .._counting import read_lines, row_total


def nwords_counter(nwords_list: TextIO, n: int = 4, end_chr: str = "\x03", threshold: int = 10):
    reach = n - 1
    words: Dict[str, int] = {}
    uniq: Dict[tuple, int] = {}
    for line in read_lines(nwords_list):
        fields = line.strip("\r\n").split("\t")
        text = fields[0] + end_chr
        cut, pieces = 0, []
        for seg in fields[1::2] + [end_chr]:
            piece = text[cut:cut + len(seg)]
            cut += len(seg)
            pieces.append(piece)
            words[piece] = words.get(piece, 0) + 1
        if "".join(pieces) != text or len(text) < 4:
            raise Exception("error1")
        key = tuple(pieces)
        uniq[key] = uniq.get(key, 0) + 1
    frequent = {w for w, c in words.items() if c >= threshold}
    counts: Dict[str, Dict[str, int]] = {}
    for pieces, mult in uniq.items():
        ctx = ""
        for piece in pieces:
            for tok in ((piece,) if piece in frequent else piece):
                row = counts.setdefault(ctx, {})
                row[tok] = row.get(tok, 0) + mult
                ctx = (ctx + tok)[-reach:]
    table = {}
    for ctx, row in counts.items():
        total = row_total(row)
        table[ctx] = {tok: c / total for tok, c in row.items()}
    return table, words
