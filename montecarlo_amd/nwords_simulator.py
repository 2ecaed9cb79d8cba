"""Fixed-order n-gram simulator: the reference's ``NWordsMonteCarlo``
(nwords_simulator.py:13-116) with its constructor and public attributes; the
per-sample / per-password loops run as HIP kernels over the compiled table.
Training runs in the native trainer (csrc/mc_train.cpp) when the splitter is '' or one literal
character; a regular-expression splitter goes through the Python trainer of the same contract."""
import ctypes as C
from typing import TextIO, Union

from . import _native as nv
from .lib4mc.MonteCarloParent import MonteCarlo
from .lib4mc.ProbLib import expand_2d
from .models import TableBacked
from .nwords.nwords_trainer import nwords_counter
from .trained import Trained, one_code_point, read_text, splitter_code_point


class NWordsMonteCarlo(TableBacked, MonteCarlo):
    def __init__(self, training_set: Union[TextIO, None], n: int = 2, splitter: str = ' ', start4word: int = 0,
                 skip4word: int = 1, start_chr="\x00", end_chr: str = "\x03"):
        if training_set is None:
            return
        self.end_chr = end_chr
        self.min_len = 4
        self.default_start = tuple([start_chr] * (n - 1))
        self.start_chr = start_chr
        sp, sc, ec = splitter_code_point(splitter), one_code_point(start_chr), one_code_point(end_chr)
        if None not in (sp, sc, ec) and n >= 2:
            data = read_text(training_set).encode("utf-8", "surrogatepass")
            h = C.c_void_p()
            nv.check(nv.lib.mc_train_nwords(data, len(data), n, sp, start4word, skip4word, sc, ec, C.byref(h)))
            self._trained = Trained(h)
            self.words = self._trained.words
            self._flat = self._trained.flat(nv.KIND_NGRAM, order=n - 1, end_chr=end_chr, min_len=self.min_len)
            self._dev_model = None
            return
        nwords, words = nwords_counter(training_set, n, splitter, end_chr, start4word, skip4word, start_chr=start_chr)
        self.nwords = expand_2d(nwords)
        self.words = words
        self._attach(nv.KIND_NGRAM, order=n - 1, start_units=self.default_start, end_chr=end_chr,
                     min_len=self.min_len)

    # the reference's public table attribute; with the native trainer it is rebuilt on first read
    @property
    def nwords(self):
        if "_nwords" not in self.__dict__:
            self.__dict__["_nwords"] = self._trained.nwords
        return self.__dict__["_nwords"]

    @nwords.setter
    def nwords(self, table):
        self.__dict__["_nwords"] = table


def wrapper(argv=None):
    """The reference's command line (nwords_simulator.py:119-149); the flags live in montecarlo_amd/_cli.py."""
    from . import _cli
    return _cli.nwords_simulator(argv)


if __name__ == '__main__':
    from . import _cli
    _cli.run(wrapper)
