"""Guess-number estimation with the reference's entry points
(lib4mc/MonteCarloLib.py:9-68): the sampled -log2 p are sorted, turned into the
running sum of 1/(n*p), and a test password's -log2 p is bisected into it.

Here the sort (radix), the running sum (prefix scan) and the bisection (one lane
per test password) are HIP kernels; keys and positions stay on the GPU.  With
``group`` (a torch.distributed process group) the instance holds one shard of
the samples and queries all-reduce the per-shard partial sums.
"""
from math import ceil
from typing import List, TextIO, Tuple

import numpy
import torch

from ..engine import DeviceRank
from .MonteCarloParent import SampleKeys, gc_paused


class MonteCarloLib:
    def __init__(self, minus_log_prob_list, group=None, n_global: int = None):
        if isinstance(minus_log_prob_list, SampleKeys):
            keys = minus_log_prob_list.tensor
            if group is None and minus_log_prob_list.group is not None:     # sharded sample(): its group and n
                group, n_global = minus_log_prob_list.group, minus_log_prob_list.n_global
        elif torch.is_tensor(minus_log_prob_list):
            keys = minus_log_prob_list
        else:
            keys = torch.tensor(minus_log_prob_list, dtype=torch.float64).cuda()
        keys = keys.contiguous()
        self.__group = group
        if group is not None and n_global is None:
            n = torch.tensor([keys.numel()], dtype=torch.int64, device=keys.device)
            torch.distributed.all_reduce(n, group=group)
            n_global = int(n.item())
        self.__rank = DeviceRank(keys, n_global=n_global)
        # the reference sorts the caller's list in place (:11); keep that side effect
        if isinstance(minus_log_prob_list, SampleKeys):
            minus_log_prob_list._replace(self.__rank.keys_sorted)
        elif isinstance(minus_log_prob_list, list):
            minus_log_prob_list[:] = self.__rank.keys_sorted.cpu().numpy().tolist()
        self.__gc = None
        self.__cols = None

    @property
    def device_rank(self) -> DeviceRank:
        return self.__rank

    def __estimates(self, xs) -> Tuple[numpy.ndarray, numpy.ndarray]:
        """positions[idx-1] (0 where idx == 0) and idx for every x, summed over shards."""
        x = torch.as_tensor(numpy.asarray(xs, dtype=numpy.float64)).to(self.__rank.tdev)
        pos, cnt = self.__rank.query(x.contiguous())
        if self.__group is not None:
            # every weight is positive: the summed position is > 0 exactly when some shard counted a sample
            torch.distributed.all_reduce(pos, group=self.__group)
            cnt = (pos > 0).to(torch.int64)
        pos_h = pos.cpu().numpy()
        if numpy.isnan(pos_h).any() and not numpy.isnan(numpy.asarray(xs, dtype=numpy.float64)).any():
            # a failed build poisons positions with NaN (on any shard: NaN survives the sum); never a guess number
            raise RuntimeError("rank estimation failed: positions contain NaN (mc_rank_build reported an error "
                               "on this device or on another shard)")
        return pos_h, cnt.cpu().numpy()

    def ml2p2rank(self, minus_log_prob):
        pos, cnt = self.__estimates([minus_log_prob])
        return pos[0] if cnt[0] > 0 else 1

    def ml2p_iter2gc(self, minus_log_prob_iter: List[Tuple[str, int, float]],
                     need_resort: bool = False, add1: bool = True) \
            -> List[Tuple[str, float, int, int, int, float]]:
        if need_resort:
            minus_log_prob_iter = sorted(minus_log_prob_iter, key=lambda x: x[2])
        items = list(minus_log_prob_iter)
        self.__cols = None
        if not items:
            self.__gc = []
            return self.__gc
        with gc_paused():
            return self.__columns(items, add1)

    def __columns(self, items, add1: bool):
        pwds, apps, mlps = [t[0] for t in items], [t[1] for t in items], [t[2] for t in items]
        pos, cnt = self.__estimates(mlps)
        est = numpy.where(cnt > 0, pos, 1.0)
        ranks = _monotone_ranks(est, cnt, add1)
        total = sum(apps)
        try:
            running = numpy.cumsum(numpy.asarray(apps, dtype=numpy.int64))
            exact = total < 2 ** 53 and running.size == len(apps)
        except (OverflowError, TypeError, ValueError):
            exact = False
        if exact:
            # c / total * 100 with both ints below 2^53: numpy's float64 division gives Python's quotient
            ratio = running / float(total) * 100
            cracked = running.tolist()
            gc = list(zip(pwds, mlps, apps, ranks, cracked, ratio.tolist()))
            # the columns write2 formats, kept while the list it returns is the one write2 sees
            self.__cols = (pwds, numpy.asarray(mlps, dtype=numpy.float64), numpy.asarray(apps, dtype=numpy.int64),
                           ranks, running, ratio)
        else:
            cracked, run = [], 0
            for a in apps:
                run += a
                cracked.append(run)
            gc = [(p, m, a, r, c, c / total * 100) for p, m, a, r, c in zip(pwds, mlps, apps, ranks, cracked)]
        self.__gc = gc
        return gc

    def write2(self, fd: TextIO):
        if not fd.writable():
            raise Exception(f"{fd.name} is not writable")
        if self.__gc is None:
            raise Exception(f"run mlps2gc before invoke this method")
        text = _format_native(self.__gc, self.__cols)
        self.__cols = None
        if text is not None:
            fd.write(text)
        else:
            fd.writelines(f"{pwd}\t{mlp:.8f}\t{appearance}\t{rank}\t{cracked}\t{cracked_ratio:5.2f}\n"
                          for pwd, mlp, appearance, rank, cracked, cracked_ratio in self.__gc)
        self.__gc = None


def _format_native(gc, cols=None) -> "str | None":
    """write2's lines through mc_format_gc (csrc/mc_ingest.cpp), or None when a row is not what it takes
    (a components list instead of a string, a guess number beyond int64, a newline inside a password):
    the f-string loop then writes them, as the reference does.  ``cols``: the columns ml2p_iter2gc
    kept, used when the list still is the one it returned (same length, same first / middle / last row)."""
    import ctypes as C

    from .. import _native as nv
    n = len(gc)
    if n < 512:
        return None
    fresh = cols is not None and len(cols[0]) == n and all(
        gc[i][0] is cols[0][i] and gc[i][3] == cols[3][i] and gc[i][1] == cols[1][i] for i in (0, n // 2, n - 1))
    try:
        if fresh:
            pwds, mlp, apps, ranks, crk, ratio = cols
            rank = numpy.asarray(ranks, dtype=numpy.int64)
        else:
            pwds = [row[0] for row in gc]
            rank = numpy.fromiter((row[3] for row in gc), dtype=numpy.int64, count=n)
            apps = numpy.fromiter((row[2] for row in gc), dtype=numpy.int64, count=n)
            crk = numpy.fromiter((row[4] for row in gc), dtype=numpy.int64, count=n)
            mlp = numpy.fromiter((row[1] for row in gc), dtype=numpy.float64, count=n)
            ratio = numpy.fromiter((row[5] for row in gc), dtype=numpy.float64, count=n)
        if not all(type(p) is str for p in pwds):
            return None
        blob = "\n".join(pwds).encode("utf-8")                 # (lone surrogates: not UTF-8, take the loop)
    except (UnicodeEncodeError, OverflowError, TypeError, ValueError):
        return None
    if blob.count(b"\n") != n - 1:
        return None
    if not (numpy.isfinite(mlp).all() and numpy.isfinite(ratio).all()):
        return None                                            # (C prints a NaN's sign, Python does not)
    if float(numpy.abs(mlp).max()) >= 1e18 or float(numpy.abs(ratio).max()) >= 1e9:
        return None                                            # (wider than the native writer's 120 characters)
    mlp, apps, rank, crk, ratio = (numpy.ascontiguousarray(x) for x in (mlp, apps, rank, crk, ratio))
    out = numpy.empty(len(blob) + 120 * n, dtype=numpy.uint8)           # (not zero-filled)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    size = nv.check(nv.lib.mc_format_gc(blob, len(blob), n, vp(mlp), vp(apps), vp(rank), vp(crk), vp(ratio),
                                        vp(out), out.size))
    return str(memoryview(out)[:size], "utf-8")


def _monotone_ranks(est: numpy.ndarray, cnt: numpy.ndarray, add1: bool) -> List[int]:
    """rank_i = ceil(max(est_i, rank_{i-1} + add)), rank_0 = 0 (reference :47-55).
    While every value is below 2^52 (float and int comparisons exact) this is a running
    maximum; beyond that it is evaluated with the reference's own expression, whose
    numpy-float-vs-int comparisons are themselves lossy there."""
    if est.size and numpy.all(numpy.isfinite(est)) and float(est.max()) < 2.0 ** 52:
        c = numpy.ceil(est).astype(numpy.int64)
        if add1:
            i = numpy.arange(1, c.size + 1, dtype=numpy.int64)
            s = numpy.maximum.accumulate(numpy.maximum(c - i, 0))
            return (s + i).tolist()
        return numpy.maximum.accumulate(numpy.maximum(c, 0)).tolist()
    out, prev, step = [], 0, 1 if add1 else 0
    for e, k in zip(est, cnt):
        prev = ceil(max(e if k > 0 else 1, prev + step))
        out.append(prev)
    return out
