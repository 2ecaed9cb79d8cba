"""Test-file ingest through the native line counter (csrc/mc_ingest.cpp): the unique lines of a
test set in first-seen order with their multiplicities, already as code points + offsets for the
scoring kernels (reference lib4mc/MonteCarloParent.py:56-60)."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np

from . import _native as nv

_DT = {nv.LINES_COUNT: "<u8", nv.LINES_OFF: "<u8", nv.LINES_CPS: "<u4", nv.LINES_INFO: "<u8"}


class UniqueLines:
    """count[n], off[n+1], cps[], longest; ``strings()`` materialises the Python strs."""

    def __init__(self, text: str):
        raw = text.encode("utf-8", "surrogatepass")
        h = C.c_void_p()
        nv.check(nv.lib.mc_lines_ingest(raw, len(raw), C.byref(h)))
        try:
            self.count = self._array(h, nv.LINES_COUNT)
            self.off = self._array(h, nv.LINES_OFF)
            self.cps = self._array(h, nv.LINES_CPS)
            info = self._array(h, nv.LINES_INFO)
        finally:
            nv.lib.mc_lines_free(h)
        self.n_lines, self.longest = int(info[1]), int(info[2])

    @staticmethod
    def _array(h, what: int) -> np.ndarray:
        size = nv.check(nv.lib.mc_lines_array(h, what, None, 0))
        dt = np.dtype(_DT[what])
        out = np.empty(size // dt.itemsize, dtype=dt)
        if size:
            nv.check(nv.lib.mc_lines_array(h, what, out.ctypes.data_as(C.c_void_p), size))
        return out

    def __len__(self):
        return len(self.count)

    def strings(self) -> List[str]:
        text = self.cps.tobytes().decode("utf-32-le", "surrogatepass")
        off = self.off.tolist()
        return [text[a:b] for a, b in zip(off[:-1], off[1:])]
