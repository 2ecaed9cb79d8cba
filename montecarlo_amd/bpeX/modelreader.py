"""Reader of a BPE-PCFG model directory: same contract as the reference's
bpeX/modelreader.py:15-68 (``grammar/structures.txt`` and per-tag
``<len>.txt`` terminal files)."""
import os
import re
import sys
from typing import Any, Dict, Tuple

_TAG_DIRS = (("lower", "L"), ("upper", "U"), ("mixed_2", "DM"), ("mixed_3", "TM"), ("mixed_4", "FM"),
             ("digits", "D"), ("special", "S"))
_PIECE = re.compile(r"([A-Z]+)([0-9]+)")


def check_path_exists(_path: str):
    if not os.path.exists(_path):
        print(f"{_path} not exists, exit.", file=sys.stderr)
        sys.exit(-1)


def read_tag(tag_path: str, tag: str) -> Dict[Tuple[str, int], Dict[str, float]]:
    check_path_exists(tag_path)
    rows: Dict[Tuple[str, int], Dict[str, float]] = {}
    for root, _dirs, files in os.walk(tag_path):
        for name in files:
            key = (tag, int(name[:name.find(".")]))
            with open(os.path.join(root, name)) as fd:
                for line in fd:
                    term, prob = line.strip("\r\n").split("\t")
                    rows.setdefault(key, {})[term] = float(prob)
    return rows


def read_grammars(gram_path: str):
    grammars = {}
    with open(gram_path) as fd:
        for line in fd:
            raw, prob = line.strip("\r\n").split("\t")
            grammars[tuple((m.group(1), int(m.group(2))) for m in _PIECE.finditer(raw))] = float(prob)
    return grammars


def read_bpe(model_path: str) -> Tuple[Dict[Any, float], Dict[Tuple[str, int], Dict[Any, float]]]:
    check_path_exists(model_path)
    grammars = read_grammars(os.path.join(model_path, "grammar", "structures.txt"))
    terminals: Dict[Tuple[str, int], Dict[Any, float]] = {}
    for sub, tag in _TAG_DIRS:
        terminals.update(read_tag(os.path.join(model_path, sub), tag))
    return grammars, terminals
