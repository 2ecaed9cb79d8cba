"""Secondary (incrementally trained) backoff simulator: the reference's
``BackWordsSecondaryMonteCarlo`` (backwords_secondary_simulator.py:11-25).  The
model is the ``(counts, words, config)`` tuple the secondary trainer produces,
or a binary file holding its pickle."""
import pickle

from .backwords.backwords_secondary_trainer import freq2prob
from .backwords_simulator import BackWordsMonteCarlo
from .lib4mc.ProbLib import expand_2d


class BackWordsSecondaryMonteCarlo(BackWordsMonteCarlo):
    def __init__(self, model, max_iter: int = 10 ** 100):
        super().__init__(None)
        counts, words, config = model if isinstance(model, tuple) else pickle.load(model)
        table = expand_2d(freq2prob(counts, config['threshold']))
        self._adopt(table, words, config['start_chr'], config['end_chr'], max_iter)
