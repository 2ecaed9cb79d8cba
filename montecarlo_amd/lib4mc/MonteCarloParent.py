"""The plugin base class, with the reference's surface
(lib4mc/MonteCarloParent.py:11-69): ``minus_log2``, abstract ``calc_ml2p`` /
``sample1``, batch ``sample`` and ``parse_file``.

What changed is the inside: a subclass that attaches a compiled table
(``self._flat`` / ``self._dev``, see montecarlo_amd.models) gets ``sample`` and
``parse_file`` executed as HIP kernels over the whole batch.  A third-party
subclass that only overrides ``sample1`` / ``calc_ml2p`` still works through the
per-item loops, exactly like the reference -- that is the plugin's own code, not
a fallback of the engine.
"""
import abc
import collections.abc
import contextlib
import gc
import random
from math import log2
from typing import Dict, List, TextIO, Tuple, Union

import numpy as np
import torch

from .FileLib import wc_l


@contextlib.contextmanager
def gc_paused():
    """Building 10^6 result tuples triggers the cycle collector again and again, each time walking every
    object built so far (measured: 0.65 -> 0.50 s for the columns of ml2p_iter2gc, 3x with zip(*rows));
    none of these rows can form a cycle."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class SampleKeys(collections.abc.Sequence):
    """The list ``sample`` returns: -log2 p of every sample in draw order, kept on
    the GPU.  It behaves as a read-only sequence of floats (materialised on first
    element access); ``MonteCarloLib`` consumes it without a host round trip.
    10^7..10^10 keys cannot be a Python list."""

    def __init__(self, tensor, group=None, n_global: int = None):
        self._tensor, self._later, self._len = tensor, None, int(tensor.numel())
        self._host = None
        # sharded sampling (montecarlo_amd.dist.use_group): this rank's share of n_global keys
        self.group, self.n_global = group, n_global

    @property
    def tensor(self):
        if self._later is not None:                          # (the sorted keys, written when somebody looks)
            self._tensor, self._later = self._later(), None
        return self._tensor

    def __len__(self):
        return self._len

    def _materialise(self):
        if self._host is None:
            self._host = self.tensor.cpu().numpy()
        return self._host

    def __getitem__(self, i):
        v = self._materialise()[i]
        return v.tolist() if isinstance(i, slice) else float(v)

    def __iter__(self):
        return iter(self._materialise().tolist())

    def tolist(self) -> List[float]:
        return self._materialise().tolist()

    def _replace(self, tensor):
        """``tensor``: the new keys, or a function that returns them (MonteCarloLib sorts this list "in place";
        with the run-length index the sorted array only exists once it is asked for)."""
        if callable(tensor):
            self._tensor, self._later, self._host = None, tensor, None
        else:
            self._tensor, self._later, self._host = tensor, None, None


class MonteCarlo(metaclass=abc.ABCMeta):
    _flat = None         # montecarlo_amd.tables.FlatTable once a subclass attaches a compiled table
    _dev_model = None    # montecarlo_amd.engine.DeviceModel, uploaded on first use
    _seed = None
    _next_id = 0
    max_draws = 1 << 16  # bound on uniforms per sample; the reference has none (it would spin)

    @staticmethod
    def minus_log2(prob: float) -> float:
        return -log2(prob)

    @abc.abstractmethod
    def calc_ml2p(self, pwd: str) -> Tuple[float, List[str]]:
        return .0, []

    @abc.abstractmethod
    def sample1(self) -> (float, str):
        return .0, ""

    @property
    def _dev(self):
        if self._dev_model is None:
            from ..engine import DeviceModel
            self._dev_model = DeviceModel(self._flat)     # raises without a GPU: no CPU fallback
        return self._dev_model

    # ---- Philox stream bookkeeping -----------------------------------------------------------
    def seed(self, seed: int, first_id: int = 0):
        """Fix the counter-based stream: sample ``i`` of this model always uses
        uniforms (seed, i, 0..).  Unset, the seed is drawn from ``random`` so that
        ``random.seed(...)`` makes a run reproducible, as with the reference."""
        self._seed, self._next_id = int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_id)

    def _claim_ids(self, n: int) -> Tuple[int, int]:
        if self._seed is None:
            self._seed = random.getrandbits(64)
        first = self._next_id
        self._next_id += n
        return self._seed, first

    # ---- batch sampling (reference :33-46) -----------------------------------------------------
    def sample(self, size: int, sampled_pwds: Dict[str, List] = None, clearIfNotNone: bool = True):
        if self._flat is None:
            return self._sample_by_loop(size, sampled_pwds, clearIfNotNone)
        seed, first = self._claim_ids(size)
        want = isinstance(sampled_pwds, dict)
        stride = self._sample_stride() if want else 0
        from .. import dist as mcd
        group, n_global = mcd.active_group(), size
        if group is not None:
            # this rank's contiguous share of the SAME sample ids a single GPU would draw
            lo, size = mcd.shard_range(n_global, torch.distributed.get_rank(group),
                                       torch.distributed.get_world_size(group))
            first += lo
        res = self._dev.sample(size, seed, first_id=first, max_draws=self.max_draws, sym_stride=stride)
        self._dev.check()
        if want:
            seen: Dict[str, List] = {}
            distinct = self._distinct_strings(res, size, seed, first) if size else []
            if group is not None:
                distinct = mcd.merge_distinct(distinct, group)
            for pwd, x, c in distinct:
                hit = seen.get(pwd)
                if hit is None:
                    seen[pwd] = [x, c]
                else:
                    hit[1] += c
            if clearIfNotNone:
                sampled_pwds.clear()
            sampled_pwds.update(seen)
        return SampleKeys(res["ml2p"], group=group, n_global=n_global if group is not None else None)

    def _sample_stride(self) -> int:
        """Symbols kept per sample when strings are wanted (longer ones are replayed one by one)."""
        return 32

    def _decode_rows(self, syms: np.ndarray, lens: np.ndarray) -> List[str]:
        drop = getattr(self, "start_chr", "") or ""
        return self._flat.decode(syms, lens, drop=drop)

    def _distinct_strings(self, res, size: int, seed: int, first: int):
        """(string, ml2p of its first occurrence, occurrences) in first-sampled order: the rows are
        deduplicated on the GPU (mc_sample_dedup), only one row per distinct string comes to the host."""
        stride = res["syms"].shape[1]
        idx, cnt, long_ids = self._dev.dedup(res["syms"], res["len"], size)
        rows = res["syms"][idx].cpu().numpy()
        lens = res["len"][idx].cpu().numpy()
        ml2p = res["ml2p"][idx].cpu().numpy().tolist()
        recs = list(zip(idx.cpu().numpy().tolist(), self._decode_rows(rows, lens), ml2p, cnt.cpu().numpy().tolist()))
        for i in long_ids.cpu().numpy().tolist():       # rare long strings: replay that sample id alone
            n = int(res["len"][i])
            one = self._dev.sample(1, seed, first_id=first + i, max_draws=self.max_draws, sym_stride=n)
            pwd = self._decode_rows(one["syms"].cpu().numpy(), one["len"].cpu().numpy())[0]
            recs.append((i, pwd, float(one["ml2p"].cpu()[0]), 1))
        if len(long_ids):
            recs.sort(key=lambda r: r[0])
        return [(p, x, c) for _, p, x, c in recs]

    def _strings_of(self, res, seed: int, first: int) -> List[str]:
        lens = res["len"].cpu().numpy()
        syms = res["syms"].cpu().numpy()
        stride = syms.shape[1]
        out = self._decode_rows(syms, np.minimum(lens, stride))
        for i in np.nonzero(lens > stride)[0]:       # rare long strings: replay that sample id alone
            one = self._dev.sample(1, seed, first_id=first + int(i), max_draws=self.max_draws,
                                   sym_stride=int(lens[i]))
            out[i] = self._decode_rows(one["syms"].cpu().numpy(), one["len"].cpu().numpy())[0]
        return out

    def _sample_by_loop(self, size, sampled_pwds, clearIfNotNone):
        results, seen = [], {}
        for _ in range(size):
            prob, pwd = self.sample1()
            results.append(prob)
            seen.setdefault(pwd, [prob, 0])[1] += 1
        if isinstance(sampled_pwds, dict):
            if clearIfNotNone:
                sampled_pwds.clear()
            sampled_pwds.update(seen)
        return results

    # ---- test-set scoring (reference :48-69) ---------------------------------------------------
    def score_batch(self, pwds: List[str]):
        """(ml2p list, components list) for many passwords; models with a device table
        override this with one kernel launch."""
        pairs = [self.calc_ml2p(p) for p in pwds]
        return [m for m, _ in pairs], [c for _, c in pairs]

    def _score_unique(self, lines, pwds: List[str], want_parts: bool):
        return self.score_batch(pwds)

    def _score_unique_device(self, lines):
        """-log2 p of ``lines`` (montecarlo_amd.ingest.UniqueLines) as a device float64 vector, or None when
        the model scores through ``score_batch`` only (a third-party subclass)."""
        return None

    def parse_file(self, testing_set: TextIO, using_component: bool = False) \
            -> List[Tuple[Union[str, List[str]], int, float]]:
        if self._flat is not None and testing_set.seekable():
            testing_set.seek(0)                             # wc_l of the reference rewinds (:56)
            text = testing_set.read()
            if "\r" not in text:
                # every newline mode of a text file splits such a text at '\n' only: the native line
                # counter sees what iterating the file would (a text with '\r' in it goes below)
                from ..ingest import UniqueLines
                from ..columns import ScoredList
                lines = UniqueLines(text)
                keys = self._score_unique_device(lines) if not using_component and len(lines) else None
                if keys is not None:
                    # scores stay on the GPU until they are sorted (mc_order_keys: Python's stable sort by
                    # ml2p); the passwords stay code points until somebody reads a row
                    from ..engine import device_order
                    ml2p_d, order_d = device_order(keys.contiguous())
                    order = order_d.cpu().numpy().view(np.uint32)
                    return ScoredList(order, lines.count[order].astype(np.int64), ml2p_d.cpu().numpy(), lines=lines)
                pwds = lines.strings()
                ml2p, parts = self._score_unique(lines, pwds, using_component)
                ml2p = np.asarray(ml2p, dtype=np.float64)
                order = np.argsort(ml2p, kind="stable")
                first = parts if using_component else pwds
                return ScoredList(order.astype(np.uint32), lines.count[order].astype(np.int64), ml2p[order],
                                  first=[first[i] for i in order.tolist()], passwords=not using_component)
        wc_l(testing_set)                                   # rewinds, as the reference does
        counts: Dict[str, int] = {}
        for line in testing_set:
            line = line.strip("\r\n")
            counts[line] = counts.get(line, 0) + 1
        pwds = list(counts)
        ml2p, parts = self.score_batch(pwds)
        order = np.argsort(np.asarray(ml2p, dtype=np.float64), kind="stable")
        first = parts if using_component else pwds
        return [(first[i], counts[pwds[i]], ml2p[i]) for i in order.tolist()]
