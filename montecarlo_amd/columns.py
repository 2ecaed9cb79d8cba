"""The lists ``parse_file`` and ``ml2p_iter2gc`` return, kept as columns.

The reference returns plain lists of tuples (lib4mc/MonteCarloParent.py:68, lib4mc/MonteCarloLib.py:56-58).  At
10^6 test passwords building those tuples costs more than everything the GPU does, and a driver that only hands
the list on (``ml2p_iter2gc(parse_file(...))`` then ``write2``: backwords_simulator.py:90-94) never looks at
them.  These classes behave as the lists they replace -- length, indexing, slicing, iteration, comparison with a
list, item assignment -- and build the tuples when somebody looks; until a caller CHANGES one, ``pristine`` says
the columns still are the truth and the next stage (``ml2p_iter2gc``, ``write2``) reads them directly.

Where they are NOT a ``list``: ``isinstance(x, list)`` is False, so ``json.dumps(x)`` refuses them (pass
``list(x)``, or ``default=list``); ``list + x`` and ``x + list`` both give a plain list of the rows; pickling and
``copy.copy`` give a plain list of the rows too (the columns reference GPU-side objects that do not travel).
"""
from __future__ import annotations

import collections.abc
from typing import List, Optional

import numpy as np


class _ColumnList(collections.abc.MutableSequence):
    def __init__(self):
        self._rows: Optional[list] = None
        self.pristine = True

    def _build(self) -> list:
        raise NotImplementedError

    def _materialise(self) -> list:
        if self._rows is None:
            from .lib4mc.MonteCarloParent import gc_paused
            with gc_paused():
                self._rows = self._build()
        return self._rows

    def __len__(self):
        return self._n if self._rows is None else len(self._rows)

    def __getitem__(self, i):
        return self._materialise()[i]

    def __iter__(self):
        return iter(self._materialise())

    def __setitem__(self, i, v):
        self._materialise()[i] = v
        self.pristine = False

    def __delitem__(self, i):
        del self._materialise()[i]
        self.pristine = False

    def insert(self, i, v):
        self._materialise().insert(i, v)
        self.pristine = False

    def sort(self, *args, **kwargs):
        self._materialise().sort(*args, **kwargs)
        self.pristine = False

    def __eq__(self, other):
        if isinstance(other, (list, _ColumnList)):
            return self._materialise() == (other._materialise() if isinstance(other, _ColumnList) else other)
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __add__(self, other):
        return self._materialise() + list(other)

    def __radd__(self, other):
        return list(other) + self._materialise()

    def __reduce__(self):                 # pickle / copy: the rows as a plain list
        return (list, (self._materialise(),))

    def __repr__(self):
        return repr(self._materialise())


class ScoredList(_ColumnList):
    """``[(password or components, appearance, ml2p)]`` ascending by ml2p (stable): parse_file's result.

    Columns, all in sorted order: ``count`` (int64), ``ml2p`` (float64), ``order`` (uint32: row -> unique line),
    ``lines`` (montecarlo_amd.ingest.UniqueLines, the passwords as code points) or ``first`` (the first column as
    a list already)."""

    def __init__(self, order: np.ndarray, count: np.ndarray, ml2p: np.ndarray, lines=None, first: List = None,
                 passwords: bool = True):
        super().__init__()
        self.order, self.count, self.ml2p = order, count, ml2p
        self.lines, self._first = lines, first
        self.passwords = passwords            # the first column holds the password strings (not component lists)
        self._n = int(len(ml2p))

    def first_column(self) -> List:
        if self._first is None:
            pwds = self.lines.strings()
            self._first = [pwds[i] for i in self.order.tolist()]
        return self._first

    def _build(self) -> list:
        return list(zip(self.first_column(), self.count.tolist(), self.ml2p.tolist()))


class GcList(_ColumnList):
    """``[(password, ml2p, appearance, guess number, cracked, cracked %)]``: ml2p_iter2gc's result.  ``ranks`` is
    an int64 array, or a list of Python ints when a guess number does not fit one."""

    def __init__(self, scored: ScoredList, ranks, cracked: np.ndarray, ratio: np.ndarray):
        super().__init__()
        self.scored, self.ranks, self.cracked, self.ratio = scored, ranks, cracked, ratio
        self._n = int(len(cracked))

    def _build(self) -> list:
        s = self.scored
        ranks = self.ranks.tolist() if isinstance(self.ranks, np.ndarray) else self.ranks
        return list(zip(s.first_column(), s.ml2p.tolist(), s.count.tolist(), ranks, self.cracked.tolist(),
                        self.ratio.tolist()))
