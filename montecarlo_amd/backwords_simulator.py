"""Backoff simulator: the reference's ``BackWordsMonteCarlo``
(backwords_simulator.py:11-49)."""
import ctypes as C
from typing import TextIO, Union

from . import _native as nv
from .backwords.backwords_trainer import backwords_counter
from .lib4mc.ProbLib import expand_2d
from .nwords_simulator import NWordsMonteCarlo
from .trained import Trained, one_code_point, read_text, splitter_code_point


class BackWordsMonteCarlo(NWordsMonteCarlo):
    def __init__(self, training_set: Union[TextIO, None], splitter: str = '', start4word: int = 0, skip4word: int = 1,
                 threshold: int = 10, start_chr: str = '\x00', end_chr: str = "\x03", max_gram: int = 256,
                 max_iter: int = 10 ** 100):
        super().__init__(None)
        if training_set is None:
            return
        sp, sc, ec = splitter_code_point(splitter), one_code_point(start_chr), one_code_point(end_chr)
        if None not in (sp, sc, ec):
            # native trainer (csrc/mc_train.cpp): same rows, no Python dict on the way to the GPU table
            data = read_text(training_set).encode("utf-8", "surrogatepass")
            h = C.c_void_p()
            nv.check(nv.lib.mc_train_backwords(data, len(data), sp, start4word, skip4word, threshold,
                                               min(max_gram, 0xFFFFFFFF), sc, ec, C.byref(h)))
            self._trained = Trained(h)
            self.end_chr, self.words, self.min_len = end_chr, self._trained.words, 4
            self.default_start = self.start_chr = start_chr
            self.max_iter = max_iter
            self._flat = self._trained.flat(nv.KIND_BACKOFF, order=0, end_chr=end_chr, min_len=self.min_len)
            self._dev_model = None
            return
        backwords, words = backwords_counter(training_set, splitter, start_chr, end_chr, start4word, skip4word,
                                             threshold=threshold, max_gram=max_gram)
        self._adopt(expand_2d(backwords), words, start_chr, end_chr, max_iter)

    def _adopt(self, table, words, start_chr, end_chr, max_iter):
        self.nwords = table
        self.end_chr = end_chr
        self.words = words
        self.min_len = 4
        self.default_start = start_chr
        self.start_chr = start_chr
        self.max_iter = max_iter
        # the history starts as list(default_start): the characters of the start string
        self._attach(nv.KIND_BACKOFF, order=0, start_units=tuple(start_chr), end_chr=end_chr, min_len=self.min_len)


def wrapper(argv=None):
    """The reference's command line (backwords_simulator.py:52-99); the flags live in montecarlo_amd/_cli.py."""
    from . import _cli
    return _cli.backwords_simulator(argv)


if __name__ == '__main__':
    from . import _cli
    _cli.run(wrapper)
