"""Host-side probability-table helpers with the reference's contracts
(lib4mc/ProbLib.py:10-43).  `expand_*` produce the (value dict, key list, fp64
cumulative sum) triples that montecarlo_amd.tables flattens for the GPU;
`pick_expand` is kept for plugins that draw on the host -- the engine's own
sampling never calls it (it runs in the HIP kernel)."""
import bisect
import random
from math import log2
from typing import Any, Dict, List, Tuple

import numpy


def expand_1d(one_d_dict: Dict[Any, float], minus_log_based: bool = False) \
        -> Tuple[Dict[Any, float], List[Any], numpy.ndarray]:
    keys = list(one_d_dict)
    cum_sums = numpy.cumsum(numpy.fromiter((one_d_dict[k] for k in keys), dtype=numpy.float64, count=len(keys)))
    values = {k: -log2(v) for k, v in one_d_dict.items()} if minus_log_based else one_d_dict
    return values, keys, cum_sums


def expand_2d(two_d_dict: Dict[Any, Dict[Any, float]], minus_log_based: bool = False) \
        -> Dict[Any, Tuple[Dict[Any, float], List[Any], numpy.ndarray]]:
    return {ctx: expand_1d(row, minus_log_based=minus_log_based) for ctx, row in two_d_dict.items() if len(row) > 0}


def pick_expand(expanded: Tuple[Dict[str, float], List[str], numpy.ndarray]) -> Tuple[float, str]:
    values, keys, cum_sums = expanded
    u = random.uniform(0, cum_sums[-1])
    k = keys[min(bisect.bisect_right(cum_sums, u), len(keys) - 1)]
    return values.get(k), k
