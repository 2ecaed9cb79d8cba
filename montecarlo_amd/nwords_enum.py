"""Enumerate every string an n-gram model gives more than a probability threshold.
The reference's repo-root script nwords_enum.py:7-39; the flags live in montecarlo_amd/_cli.py."""


def wrapper(argv=None):
    from . import _cli
    return _cli.nwords_enum(argv)


if __name__ == '__main__':
    from . import _cli
    _cli.run(wrapper)
