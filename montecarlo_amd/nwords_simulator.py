"""Fixed-order n-gram simulator: the reference's ``NWordsMonteCarlo``
(nwords_simulator.py:13-116) with its constructor and public attributes; the
per-sample / per-password loops run as HIP kernels over the compiled table."""
from typing import TextIO, Union

from . import _native as nv
from .lib4mc.MonteCarloParent import MonteCarlo
from .lib4mc.ProbLib import expand_2d
from .models import TableBacked
from .nwords.nwords_trainer import nwords_counter


class NWordsMonteCarlo(TableBacked, MonteCarlo):
    def __init__(self, training_set: Union[TextIO, None], n: int = 2, splitter: str = ' ', start4word: int = 0,
                 skip4word: int = 1, start_chr="\x00", end_chr: str = "\x03"):
        if training_set is None:
            return
        nwords, words = nwords_counter(training_set, n, splitter, end_chr, start4word, skip4word, start_chr=start_chr)
        self.nwords = expand_2d(nwords)
        self.words = words
        self.end_chr = end_chr
        self.min_len = 4
        self.default_start = tuple([start_chr] * (n - 1))
        self.start_chr = start_chr
        self._attach(nv.KIND_NGRAM, order=n - 1, start_units=self.default_start, end_chr=end_chr,
                     min_len=self.min_len)
