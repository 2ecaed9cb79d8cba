"""Train, or continue training, a secondary backoff model file.
The reference's repo-root script backwords_secondary_trainer.py:8-61; the flags live in montecarlo_amd/_cli.py."""


def wrapper(argv=None):
    from . import _cli
    return _cli.backwords_secondary_trainer(argv)


if __name__ == '__main__':
    from . import _cli
    _cli.run(wrapper)
