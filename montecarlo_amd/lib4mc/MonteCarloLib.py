"""Guess-number estimation with the reference's entry points
(lib4mc/MonteCarloLib.py:9-68): the sampled -log2 p are sorted, turned into the
running sum of 1/(n*p), and a test password's -log2 p is bisected into it.

Here the index (one entry per distinct sampled value: the value, the samples at or
below it, their summed weights -- all that bisect_right ever reads; or, for keys
that do not repeat, the sorted keys and their prefix sums) and the bisection (one
lane per test password) are HIP kernels; keys and positions stay on the GPU.  With
``group`` (a torch.distributed process group) the instance holds one shard of
the samples and queries all-reduce the per-shard partial sums.
"""
from math import ceil
from typing import List, TextIO, Tuple

import numpy
import torch

from ..engine import DeviceRank
from .MonteCarloParent import SampleKeys


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
            rank = self.__rank
            minus_log_prob_list._replace(lambda: rank.keys_sorted)
        elif isinstance(minus_log_prob_list, list):
            minus_log_prob_list[:] = self.__rank.keys_sorted.cpu().numpy().tolist()
        self.__gc = None

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
        """Reference :34-58.  The bisection, the monotone guess numbers and the running crack counts are computed
        on columns (the first two on the GPU: mc_rank_query, mc_guess_numbers); the result behaves as the list of
        tuples the reference returns and builds them only when looked at (montecarlo_amd.columns.GcList)."""
        from ..columns import GcList, ScoredList
        scored = minus_log_prob_iter
        if not (isinstance(scored, ScoredList) and scored.pristine):
            items = list(minus_log_prob_iter)
            if need_resort:
                items = sorted(items, key=lambda x: x[2])
            n = len(items)
            scored = ScoredList(order=None, count=_int_column([t[1] for t in items]),
                                ml2p=numpy.fromiter((t[2] for t in items), dtype=numpy.float64, count=n),
                                first=[t[0] for t in items],
                                passwords=all(type(t[0]) is str for t in items[:1]))
        # (a pristine parse_file result is in ascending order already: a stable resort would change nothing)
        if len(scored) == 0:
            self.__gc = []
            return self.__gc
        ranks = self.__guess_numbers(scored.ml2p, add1)
        apps = scored.count
        if apps.dtype == object:
            cracked, run = [], 0
            for a in apps.tolist():
                run += a
                cracked.append(run)
            total = run
            cracked = numpy.array(cracked, dtype=object)
            ratio = numpy.array([c / total * 100 for c in cracked.tolist()], dtype=numpy.float64)
        else:
            cracked = numpy.cumsum(apps, dtype=numpy.int64)
            total = int(cracked[-1])
            if total < 2 ** 53:
                # c / total * 100 with both ints below 2^53: numpy's float64 division gives Python's quotient.
                # (Two statements on purpose: `cracked / t * 100` lets numpy try to reuse the temporary, for which
                # it walks the C stack -- the first such walk in a process with the ROCm libraries loaded costs 80 ms.)
                ratio = cracked / float(total)
                numpy.multiply(ratio, 100, out=ratio)
            else:
                ratio = numpy.array([c / total * 100 for c in cracked.tolist()], dtype=numpy.float64)
        self.__gc = GcList(scored, ranks, cracked, ratio)
        return self.__gc

    def __guess_numbers(self, mlps: numpy.ndarray, add1: bool):
        """rank_i = ceil(max(estimate_i, rank_{i-1} + add)), rank_0 = 0 (reference :47-55), in the order given:
        int64 array from the device (mc_guess_numbers), or a list of Python ints when some guess number
        exceeds int64 -- those, a suffix in ascending order, are finished with the reference's own expression."""
        from ..engine import device_guess_numbers, device_query_guess_numbers
        rk = self.__rank
        x = torch.from_numpy(numpy.ascontiguousarray(mlps, dtype=numpy.float64)).to(rk.tdev)
        if self.__group is None:
            # (one GPU: the bisection inside the loop, as the reference writes it -- mc_rank_runs_guess_numbers)
            _pos, _cnt, ranks_d, exact_d = device_query_guess_numbers(rk, x, add1)
        else:
            pos, _cnt = rk.query(x)
            # every weight is positive: the summed position is > 0 exactly when some shard counted a sample
            torch.distributed.all_reduce(pos, group=self.__group)
            ranks_d, exact_d, _ws = device_guess_numbers(pos, None, add1)
        ranks = ranks_d.cpu().numpy()
        exact = int(exact_d.item())
        if exact >= len(mlps):
            return ranks
        # what follows does not fit int64 (or is NaN): the kernel left the estimates there, as fp64 bit patterns
        est = ranks[exact:].view(numpy.float64)
        if numpy.isnan(est).any() and not numpy.isnan(mlps).any():
            # a failed build poisons positions with NaN (on any shard: NaN survives the sum); never a guess number
            raise RuntimeError("rank estimation failed: positions contain NaN (mc_rank_build reported an error "
                               "on this device or on another shard)")
        out = ranks[:exact].tolist()
        prev, step = (out[-1] if out else 0), (1 if add1 else 0)
        for e in est.tolist():
            prev = ceil(max(e, prev + step))
            out.append(prev)
        return out

    def write2(self, fd: TextIO):
        if not fd.writable():
            raise Exception(f"{fd.name} is not writable")
        if self.__gc is None:
            raise Exception(f"run mlps2gc before invoke this method")
        text = _format_native(self.__gc)
        if text is not None:
            fd.write(text)
        else:
            fd.writelines(f"{pwd}\t{mlp:.8f}\t{appearance}\t{rank}\t{cracked}\t{cracked_ratio:5.2f}\n"
                          for pwd, mlp, appearance, rank, cracked, cracked_ratio in self.__gc)
        self.__gc = None


def _int_column(values) -> numpy.ndarray:
    try:
        a = numpy.asarray(values)
        if a.dtype.kind in "iu" and a.ndim == 1:
            return a.astype(numpy.int64)
    except (OverflowError, TypeError, ValueError):
        pass
    return numpy.array(values, dtype=object)


def _format_native(gc) -> "str | None":
    """write2's lines through the native writers (csrc/mc_ingest.cpp), or None when a row is not what they take
    (a components list instead of a string, a guess number beyond int64, a newline or a lone surrogate inside a
    password, NaN): the f-string loop then writes them, as the reference does.  A pristine GcList is formatted
    from its columns; when its passwords still are the code points of the ingest (parse_file's result went
    straight through), no Python string is built for them at all."""
    import ctypes as C

    from .. import _native as nv
    from ..columns import GcList
    n = len(gc)
    if n < 512:
        return None
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    cols = isinstance(gc, GcList) and gc.pristine and gc.scored.pristine and isinstance(gc.ranks, numpy.ndarray) \
        and gc.scored.passwords and gc.scored.count.dtype != object and gc.cracked.dtype != object
    try:
        if cols:
            sc = gc.scored
            mlp, apps, rank, crk, ratio = sc.ml2p, sc.count, gc.ranks, gc.cracked, gc.ratio
        else:
            pwds = [row[0] for row in gc]
            rank = numpy.fromiter((row[3] for row in gc), dtype=numpy.int64, count=n)
            apps = numpy.fromiter((row[2] for row in gc), dtype=numpy.int64, count=n)
            crk = numpy.fromiter((row[4] for row in gc), dtype=numpy.int64, count=n)
            mlp = numpy.fromiter((row[1] for row in gc), dtype=numpy.float64, count=n)
            ratio = numpy.fromiter((row[5] for row in gc), dtype=numpy.float64, count=n)
    except (OverflowError, TypeError, ValueError):
        return None
    if not (numpy.isfinite(mlp).all() and numpy.isfinite(ratio).all()):
        return None                                            # (C prints a NaN's sign, Python does not)
    if float(numpy.abs(mlp).max()) >= 1e18 or float(numpy.abs(ratio).max()) >= 1e9:
        return None                                            # (wider than the native writer's 120 characters)
    mlp, apps, rank, crk, ratio = (numpy.ascontiguousarray(x) for x in (mlp, apps.astype(numpy.int64, copy=False),
                                                                         rank, crk.astype(numpy.int64, copy=False),
                                                                         ratio))
    if cols and gc.scored.lines is not None and gc.scored._first is None:
        lines = gc.scored.lines
        order = numpy.ascontiguousarray(gc.scored.order, dtype=numpy.uint32)
        out = numpy.empty(4 * len(lines.cps) + 120 * n, dtype=numpy.uint8)       # (not zero-filled)
        size = nv.lib.mc_format_gc_cps(vp(lines.cps), vp(lines.off), len(lines.off) - 1, vp(order), n, vp(mlp), vp(apps), vp(rank),
                                       vp(crk), vp(ratio), vp(out), out.size)
        if size >= 0:
            return str(memoryview(out)[:size], "utf-8")
        return None                                            # (a lone surrogate: the loop writes it as Python would)
    if cols:
        pwds = gc.scored.first_column()
    try:
        if not all(type(p) is str for p in pwds):
            return None
        blob = "\n".join(pwds).encode("utf-8")                 # (lone surrogates: not UTF-8, take the loop)
    except (UnicodeEncodeError, TypeError):
        return None
    if blob.count(b"\n") != n - 1:
        return None
    out = numpy.empty(len(blob) + 120 * n, dtype=numpy.uint8)           # (not zero-filled)
    size = nv.check(nv.lib.mc_format_gc(blob, len(blob), n, vp(mlp), vp(apps), vp(rank), vp(crk), vp(ratio),
                                        vp(out), out.size))
    return str(memoryview(out)[:size], "utf-8")
