"""Secondary (incrementally trained) backoff simulator: the reference's
``BackWordsSecondaryMonteCarlo`` (backwords_secondary_simulator.py:11-25).  The
model is the ``(counts, words, config)`` tuple the secondary trainer produces,
or a binary file holding its pickle."""
import pickle

from . import _native as nv
from .backwords.backwords_secondary_trainer import freq2prob
from .trained import NativeCounts
from .backwords_simulator import BackWordsMonteCarlo
from .lib4mc.ProbLib import expand_2d


class BackWordsSecondaryMonteCarlo(BackWordsMonteCarlo):
    def __init__(self, model, max_iter: int = 10 ** 100):
        super().__init__(None)
        counts, words, config = model if isinstance(model, tuple) else pickle.load(model)
        if not isinstance(counts, NativeCounts):
            # plain dicts (a pickle written by the reference or by this package): native when well-formed
            counts = NativeCounts.from_plain(counts, words, config['start_chr'], config['end_chr']) or counts
        if isinstance(counts, NativeCounts):
            # freq2prob and the table compile in C++ (csrc/mc_train.cpp), no Python dict in between
            self._trained = counts.train(config['threshold'])
            self.end_chr, self.words, self.min_len = config['end_chr'], words, 4
            self.default_start = self.start_chr = config['start_chr']
            self.max_iter = max_iter
            self._flat = self._trained.flat(nv.KIND_BACKOFF, order=0, end_chr=self.end_chr, min_len=self.min_len)
            self._dev_model = None
            return
        table = expand_2d(freq2prob(counts, config['threshold']))
        self._adopt(table, words, config['start_chr'], config['end_chr'], max_iter)


def wrapper(argv=None):
    """The reference's command line (backwords_secondary_simulator.py:28-60); the flags live in montecarlo_amd/_cli.py."""
    from . import _cli
    return _cli.backwords_secondary_simulator(argv)


if __name__ == '__main__':
    from . import _cli
    _cli.run(wrapper)
