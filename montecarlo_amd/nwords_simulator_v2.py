"""n-words v2 simulator: the reference's ``NWords2MonteCarlo``
(nwords_simulator_v2.py:11-82).  Sampling follows the reference including its
quirk that ``pwd_len`` survives a restart (:73-81).  The reference's own
``calc_ml2p`` is defective (it returns 1022.0 for every
password of length <= 30), so there is nothing to be drop-in with; it raises."""
from typing import TextIO

from . import _native as nv
from .lib4mc.MonteCarloParent import MonteCarlo
from .lib4mc.ProbLib import expand_2d
from .models import TableBacked
from .nwords_v2.nwords_trainer_v2 import nwords_counter


class NWords2MonteCarlo(TableBacked, MonteCarlo):
    def __init__(self, training_set: TextIO, n: int, end_chr: str = "\x03"):
        nwords, words = nwords_counter(training_set, n, end_chr)
        self.nwords = expand_2d(nwords)
        self.words = words
        self.end_chr = end_chr
        self._attach(nv.KIND_V2, order=n - 1, start_units=(), end_chr=end_chr, min_len=4,
                     flags=nv.FLAG_KEEP_LEN_ON_RESTART)

    def score_batch(self, pwds):
        raise NotImplementedError("nwords_v2 scoring has no working reference (its calc_ml2p is defective)")

    def _score_unique(self, lines, pwds, want_parts):
        return self.score_batch(pwds)
