"""n-words v2 simulator: the reference's ``NWords2MonteCarlo``
(nwords_simulator_v2.py:11-82).  Sampling follows the reference including its
quirk that ``pwd_len`` survives a restart (:73-81).  The reference's own
``calc_ml2p`` is defective (it returns 1022.0 for every
password of length <= 30), so there is nothing to be drop-in with; it raises."""
import ctypes as C
from typing import TextIO

from . import _native as nv
from .lib4mc.MonteCarloParent import MonteCarlo
from .lib4mc.ProbLib import expand_2d
from .models import TableBacked
from .nwords_v2.nwords_trainer_v2 import nwords_counter
from .trained import Trained, one_code_point, read_text


class NWords2MonteCarlo(TableBacked, MonteCarlo):
    def __init__(self, training_set: TextIO, n: int, end_chr: str = "\x03"):
        self.end_chr = end_chr
        ec = one_code_point(end_chr)
        if ec is not None and n >= 2:
            # native trainer (csrc/mc_train.cpp: mc_train_v2); the table never becomes a Python dict on its
            # way to the GPU, `nwords` is rebuilt on first read
            data = read_text(training_set).encode("utf-8", "surrogatepass")
            h = C.c_void_p()
            nv.check(nv.lib.mc_train_v2(data, len(data), n, ec, 10, C.byref(h)))
            self._trained = Trained(h, v2=True)
            self.words = self._trained.words
            self._flat = self._trained.flat(nv.KIND_V2, order=n - 1, end_chr=end_chr, min_len=4,
                                            flags=nv.FLAG_KEEP_LEN_ON_RESTART)
            self._dev_model = None
            return
        nwords, words = nwords_counter(training_set, n, end_chr)
        self.nwords = expand_2d(nwords)
        self.words = words
        self._attach(nv.KIND_V2, order=n - 1, start_units=(), end_chr=end_chr, min_len=4,
                     flags=nv.FLAG_KEEP_LEN_ON_RESTART)

    # the reference's table attribute; with the native trainer it is rebuilt on first read
    @property
    def nwords(self):
        if "_nwords" not in self.__dict__:
            self.__dict__["_nwords"] = self._trained.nwords
        return self.__dict__["_nwords"]

    @nwords.setter
    def nwords(self, table):
        self.__dict__["_nwords"] = table

    def score_batch(self, pwds):
        raise NotImplementedError("nwords_v2 scoring has no working reference (its calc_ml2p is defective)")

    def _score_unique(self, lines, pwds, want_parts):
        return self.score_batch(pwds)

    def _score_unique_device(self, lines):
        return self.score_batch(None)


def main(argv=None):
    """The reference's command line (nwords_simulator_v2.py:98-114); the flags live in montecarlo_amd/_cli.py."""
    from . import _cli
    return _cli.nwords_simulator_v2(argv)


if __name__ == '__main__':
    from . import _cli
    _cli.run(main)
