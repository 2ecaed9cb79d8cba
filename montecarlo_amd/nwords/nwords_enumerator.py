"""Threshold enumerator of a fixed-order model: the reference's nwords/nwords_enumerator.py
(``minus_log2`` :5-11, ``enumerator`` :15-21, ``iterate`` :23-42) with its signature, output format and
side effect (the caller's table ends up holding -log2 p); the depth-first loop runs on the GPU as a
frontier expansion (mc_enumerate, csrc/mc_enum.hip)."""
from math import log2
from typing import Dict, TextIO, Tuple

from .. import _native as nv
from ..engine import DeviceModel
from ..lib4mc.ProbLib import expand_2d
from ..tables import FlatTable

MAX_LEN = 30        # `if cur_len > 30 ... return` (:27)


def minus_log2(nwords_dict_float: Dict[Tuple, Dict[str, float]]):
    for items in nwords_dict_float.values():
        for item, prob in items.items():
            items[item] = -log2(prob)
    return nwords_dict_float


def write_strings(flat: FlatTable, found, f_save: TextIO, digits: int):
    ml2p, lens, syms = found
    for pwd, x in zip(flat.decode(syms, lens), ml2p.tolist()):
        f_save.write(f'{pwd}\t{x:.{digits}f}\n')


def enumerator(nwords_dict_float: Dict[Tuple, Dict[str, float]], threshold: float, start_chr, end_chr, min_len,
               f_save: TextIO, order: int, max_nodes: int = 1 << 31):
    flat = FlatTable(expand_2d(nwords_dict_float), nv.KIND_NGRAM, order=order, start_units=(start_chr,) * order,
                     end_chr=end_chr, min_len=min_len)
    found = DeviceModel(flat).enumerate(threshold, min_len, MAX_LEN, max_nodes)
    minus_log2(nwords_dict_float)           # what the reference leaves in the caller's dict (:16)
    write_strings(flat, found, f_save, 8)
