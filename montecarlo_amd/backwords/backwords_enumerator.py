"""Threshold enumerator of a backoff model: the reference's backwords/backwords_enumerator.py
(``enumerator`` :15-20, ``iterate`` :22-45) with its signature, output format (5 decimals) and side
effect; the walk -- longest-suffix contexts, and an end symbol met before ``min_len`` joining the
string, as the reference does -- runs on the GPU (mc_enumerate, csrc/mc_enum.hip)."""
from typing import Dict, TextIO, Tuple

from .. import _native as nv
from ..engine import DeviceModel
from ..lib4mc.ProbLib import expand_2d
from ..nwords.nwords_enumerator import minus_log2, write_strings
from ..tables import FlatTable

MAX_LEN = 256       # `if cur_len > 256 ... return` (:33)


def enumerator(backwords_dict_float: Dict[Tuple, Dict[str, float]], threshold: float, start_chr, end_chr, min_len,
               f_save: TextIO, max_nodes: int = 1 << 31):
    flat = FlatTable(expand_2d(backwords_dict_float), nv.KIND_BACKOFF, order=0, start_units=(start_chr,),
                     end_chr=end_chr, min_len=min_len)
    found = DeviceModel(flat).enumerate(threshold, min_len, MAX_LEN, max_nodes)
    minus_log2(backwords_dict_float)
    write_strings(flat, found, f_save, 5)
