"""Command lines of the simulators, the enumerators and the stand-alone secondary trainer: the flags,
defaults and file handling of the reference's repo-root scripts (nwords_simulator.py:119-144,
backwords_simulator.py:52-90, backwords_secondary_simulator.py:28-52, bpe_simulator.py:175-199,
nwords_simulator_v2.py:98-110, nwords_enum.py:7-35, backwords_enum.py:7-20,
backwords_secondary_trainer.py:8-57), so `python -m montecarlo_amd.<script> <the same arguments>` is the
reference's command.  Every front end is a table of flags plus the five calls of the driver flow
(sample -> MonteCarloLib -> parse_file -> ml2p_iter2gc -> write2); the modules re-export theirs as
``wrapper`` / ``main`` under the reference's names and take ``argv`` so tests can call them."""
import argparse
import pickle
import sys
from typing import Callable, Optional, Sequence

SPLITTER_WORDS = {'empty': '', 'space': ' ', 'tab': '\t'}

_WORD_COLUMNS = (
    ("--start4word", dict(dest="start4word", type=int, default=0,
                          help="column of the first word after splitting a line")),
    ("--skip4word", dict(dest="skip4word", type=int, default=1,
                         help="step from one word column to the next (tags may sit between them)")),
)
_SIZE = ("--size", dict(dest="size", type=int, default=100000, help="number of Monte Carlo samples"))
_TEST = ("-t", "--test", dict(dest="test", type=argparse.FileType('r'), required=True, help="test passwords"))
_SAVE = ("-s", "--save", dict(dest="save", type=argparse.FileType('w'), required=True,
                              help="the guess-number table goes here"))
_DEBUG = ("--debug-mode", dict(dest="debug_mode", action="store_true",
                               help="read passwords from the terminal and print -log2 p of each"))
_MAX_ITER = ("--max-iter", dict(dest="max_iter", type=int, default=10 ** 20,
                                help="bound on the segmentations calc_ml2p tries"))


def _parser(title: str, *flags) -> argparse.ArgumentParser:
    cli = argparse.ArgumentParser(title)
    for *names, options in flags:
        cli.add_argument(*names, **options)
    return cli


def named_splitter(value: str) -> str:
    return SPLITTER_WORDS.get(value.lower(), value)


def driver_flow(model, test, save, size: Optional[int], **gc_options) -> None:
    """The tail every simulator's command line shares (e.g. nwords_simulator.py:139-143)."""
    from .lib4mc.MonteCarloLib import MonteCarloLib
    samples = model.sample() if size is None else model.sample(size=size)
    mc = MonteCarloLib(samples)
    mc.ml2p_iter2gc(minus_log_prob_iter=model.parse_file(test), **gc_options)
    mc.write2(save)
    if save not in (sys.stdout, sys.stderr):
        save.close()


def debug_prompt(model) -> None:
    """--debug-mode of the backoff simulators (backwords_simulator.py:82-88): the prompt ends after the
    word ``exit`` has been scored too, as there."""
    typed = ""
    while typed != "exit":
        typed = input("type in passwords: ")
        print(model.calc_ml2p(typed)[0])


def nwords_simulator(argv: Optional[Sequence[str]] = None) -> None:
    from .nwords_simulator import NWordsMonteCarlo
    args = _parser(
        "N words simulator",
        ("-i", "--input", dict(dest="input", type=argparse.FileType('r'), required=True, help="training file")),
        _TEST, _SAVE,
        ("-n", "--ngram", dict(dest="ngram", type=int, default=2, choices=[2, 3, 4, 5, 6], help="n of the n-gram")),
        _SIZE,
        ("--splitter", dict(dest="splitter", type=lambda text: str(text).replace("\\\\", "\\"), default="\t",
                            help="column separator of the training file; \"empty\" stands for ''")),
        *_WORD_COLUMNS,
    ).parse_args(argv)
    splitter = '' if args.splitter == 'empty' else args.splitter         # only this word is named here (:135)
    model = NWordsMonteCarlo(args.input, splitter=splitter, n=args.ngram, start4word=args.start4word,
                             skip4word=args.skip4word)
    driver_flow(model, args.test, args.save, args.size)


def backwords_simulator(argv: Optional[Sequence[str]] = None) -> None:
    from .backwords_simulator import BackWordsMonteCarlo
    args = _parser(
        "Backoff words simulator",
        ("-i", "--input", dict(dest="input", type=argparse.FileType('r'), required=True, help="training file")),
        _TEST, _SAVE, _SIZE,
        ("--splitter", dict(dest="splitter", type=str, default="empty",
                            help="column separator: empty | space | tab | a literal")),
        *_WORD_COLUMNS,
        ("--threshold", dict(dest="threshold", type=int, default=10, help="counts below it are pruned")),
        _DEBUG,
        ("--max-gram", dict(dest="max_gram", type=int, default=256, help="longest context")),
        _MAX_ITER,
    ).parse_args(argv)
    model = BackWordsMonteCarlo(args.input, splitter=named_splitter(args.splitter), start4word=args.start4word,
                                skip4word=args.skip4word, threshold=args.threshold, max_gram=args.max_gram,
                                max_iter=args.max_iter)
    if args.debug_mode:
        return debug_prompt(model)
    driver_flow(model, args.test, args.save, args.size)


def backwords_secondary_simulator(argv: Optional[Sequence[str]] = None) -> None:
    from .backwords_secondary_simulator import BackWordsSecondaryMonteCarlo
    args = _parser(
        "Backoff words simulator",
        ("-m", "--model", dict(dest="model", type=argparse.FileType('rb'), required=True,
                               help="model file written by the secondary trainer")),
        _TEST, _SAVE, _SIZE, _DEBUG, _MAX_ITER,
    ).parse_args(argv)
    model = BackWordsSecondaryMonteCarlo(args.model, max_iter=args.max_iter)
    if args.debug_mode:
        return debug_prompt(model)
    driver_flow(model, args.test, args.save, args.size)


def bpe_simulator(argv: Optional[Sequence[str]] = None) -> None:
    from .bpe_simulator import BpePcfgSim
    args = _parser(
        "BPE PCFG Simulator",
        ("-m", "--model", dict(dest="model", type=str, required=True, help="model directory")),
        ("-t", "--target", dict(dest="target", type=argparse.FileType('r'), required=True, help="test passwords")),
        _SAVE,
        ("--size", dict(dest="size", type=int, default=1000000, help="number of Monte Carlo samples")),
    ).parse_args(argv)
    # the only front end that asks for a re-sort and the +1 (bpe_simulator.py:168)
    driver_flow(BpePcfgSim(model_path=args.model), args.target, args.save, args.size, need_resort=True, add1=True)


def nwords_simulator_v2(argv: Optional[Sequence[str]] = None) -> None:
    from .nwords_simulator_v2 import NWords2MonteCarlo
    args = _parser(
        "NWords v2",
        ("-f", "--file", dict(dest="training", type=argparse.FileType('r'), required=True, help="training set")),
        ("-t", "--target", dict(dest="testing", type=argparse.FileType('r'), required=True, help="testing set")),
        ("-s", "--save", dict(dest="save", type=argparse.FileType('w'), default=sys.stdout, help="results")),
    ).parse_args(argv)
    # (as with the reference, scoring the test set fails: v2's calc_ml2p is defective there and raises here)
    driver_flow(NWords2MonteCarlo(args.training, 4), args.testing, args.save, None)


_CANDIDATES = (
    ("-p", "--min-prob", dict(dest="min_prob", type=float, required=True,
                              help="bound on -log2 p of a candidate (the reference compares this value with -log2 p, "
                                   "nwords_enumerator.py:27)")),
    ("-l", "--min-length", dict(dest="min_len", type=int, default=4, help="shortest candidate")),
    ("-s", "--save", dict(dest="f_save", type=argparse.FileType('w'), required=True, help="candidates go here")),
)
_PWD_FILE = ("-f", "--pwd-file", dict(dest="pwd_file", type=argparse.FileType('r'), required=True,
                                      help="training file"))


def nwords_enum(argv: Optional[Sequence[str]] = None) -> None:
    from .nwords.nwords_enumerator import enumerator
    from .nwords.nwords_trainer import nwords_counter
    args = _parser(
        "Backoff Enumerator", _PWD_FILE,
        ("-n", "--ngram", dict(dest="ngram", type=int, required=True, help="n of the n-gram")),
        ("--splitter", dict(dest="splitter", type=str, required=True, choices=sorted(SPLITTER_WORDS),
                            help="how a line splits into sub-words")),
        ("--start", dict(dest="start4words", type=int, help="column of the first sub-word")),
        ("--step", dict(dest="skip4words", type=int, help="step from one sub-word column to the next")),
        *_CANDIDATES,
    ).parse_args(argv)
    # --start / --step have no default in the reference (nwords_enum.py:15-19): left out they arrive as
    # None in the trainer's slice, which then takes every column
    table, _ = nwords_counter(args.pwd_file, n=args.ngram, splitter=SPLITTER_WORDS[args.splitter], end_chr='\x00',
                              start_chr='\x03', start4words=args.start4words, skip4words=args.skip4words)
    enumerator(table, threshold=args.min_prob, end_chr='\x00', start_chr='\x03', min_len=args.min_len,
               f_save=args.f_save, order=args.ngram - 1)
    args.f_save.close()


def backwords_enum(argv: Optional[Sequence[str]] = None) -> None:
    from .backwords.backwords_enumerator import enumerator
    from .backwords.backwords_trainer import backwords_counter
    args = _parser("Backoff Enumerator", _PWD_FILE, *_CANDIDATES).parse_args(argv)
    # positional as in backwords_enum.py:18-19: splitter '', start '\x00', end '\x03', columns 0/1,
    # threshold 10, max_gram 256; then (threshold, start, end, min_len, file)
    table, _ = backwords_counter(args.pwd_file, '', '\x00', '\x03', 0, 1, 10, 256)
    enumerator(table, args.min_prob, '\x00', '\x03', args.min_len, args.f_save)
    args.f_save.close()


def backwords_secondary_trainer(argv: Optional[Sequence[str]] = None) -> None:
    """Train (or continue training) a secondary backoff model and pickle ``(counts, words, config)``."""
    from .backwords.backwords_secondary_trainer import backwords_counter
    args = _parser(
        "Backoff: subword level trainer using secondary training file",
        ("-t", "--training", dict(dest="training", type=argparse.FileType('r'), required=True, help="training file")),
        ("-s", "--save", dict(dest="save", type=str, required=True, help="the model file to write")),
        ("-m", "--model", dict(dest="model", type=str, default=None, help="continue from this model file")),
        ("--splitter", dict(dest="splitter", type=str, default="empty",
                            help="column separator: empty | space | tab | a literal")),
        ("--start4word", dict(dest="start4words", type=int, default=0, help="column of the first word")),
        ("--skip4word", dict(dest="skip4words", type=int, default=1, help="step between word columns")),
        ("--threshold", dict(dest="threshold", type=int, default=10, help="counts below it are pruned")),
        ("--max-gram", dict(dest="max_gram", type=int, default=256, help="longest context")),
    ).parse_args(argv)
    counts, words, start_chr, end_chr, trained_on = None, None, '\x03', '\x00', []
    if args.model is not None:
        print(f"Secondary training based on: {args.model}", file=sys.stderr)
        with open(args.model, 'rb') as fin:
            # the user's own model file, written by this trainer or the reference's
            counts, words, config = pickle.load(fin)
        start_chr, end_chr, trained_on = config['start_chr'], config['end_chr'], list(config['training_list'])
        print(f"Prior training files: {','.join(trained_on)}.", file=sys.stderr)
    counts, words = backwords_counter(
        nwords_list=args.training, splitter=named_splitter(args.splitter), start_chr=start_chr, end_chr=end_chr,
        start4words=args.start4words, step4words=args.skip4words, max_gram=args.max_gram, nwords_dict=counts,
        words=words, threshold=args.threshold)
    config = {'start_chr': start_chr, 'end_chr': end_chr, 'max_gram': args.max_gram, 'threshold': args.threshold,
              'training_list': trained_on + [args.training.name]}
    with open(args.save, 'wb') as fout:
        pickle.dump((counts, words, config), file=fout)


def run(front_end: Callable[[Optional[Sequence[str]]], None]) -> None:
    """``if __name__ == '__main__'`` of every script: Ctrl-C ends with the reference's message and -1."""
    try:
        front_end(None)
    except KeyboardInterrupt:
        print("You canceled the process", file=sys.stderr)
        sys.exit(-1)
