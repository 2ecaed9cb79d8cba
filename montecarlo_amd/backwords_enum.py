"""Enumerate every string a backoff model gives more than a probability threshold.
The reference's repo-root script backwords_enum.py:7-24; the flags live in montecarlo_amd/_cli.py."""


def wrapper(argv=None):
    from . import _cli
    return _cli.backwords_enum(argv)


if __name__ == '__main__':
    from . import _cli
    _cli.run(wrapper)
