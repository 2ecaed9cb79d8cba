"""Secondary training loop (BASELINE config 5): the reference's ``backwords_secondary_main.py``
(:19-251) with the same command line, round body and output files; every round's sampling,
scoring, sort/scan and rank queries run on the GPU through the lib4mc classes of this package.

One round (``secondary_cracker``, reference :19-88): merge the round's training passwords into the
raw counts -> model -> sample `size` strings -> guess-number curve -> score the test set -> pick the
cracked passwords (``samples`` strategy: the test passwords whose STRING was sampled, guess number =
1 + samples sorted before it) -> they are the next round's training data.

Reproducibility: every round's model draws its Philox seed from ``random.getrandbits(64)`` on first
use, so ``random.seed(x)`` before ``wrapper()`` fixes the whole run (tests/golden/make_golden_secondary.py
pins it against the reference run with the same streams injected).

Multi-GPU (config 5 on 8 GPUs): launched through ``torch.distributed.run`` every rank runs this same
driver; ``montecarlo_amd.dist.use_group`` shards each round's sampling over the ranks (global sample
ids: the sample set, the cracked sets and every output file are those of a single-GPU run), the
``sampled_pwds`` dict is merged over the ranks and the guess-number queries are all-reduced; rank 0
writes the result files, the other ranks write theirs into a scratch directory."""
import argparse
import json
import math
import os
import pickle
import random
import sys
from typing import List, Tuple

from .backwords.backwords_secondary_trainer import backwords_counter
from .backwords_secondary_simulator import BackWordsSecondaryMonteCarlo
from .lib4mc.FileLib import wc_l
from .lib4mc.MonteCarloLib import MonteCarloLib

GN_CAP = 10 ** 14       # the reference's `upper_bound` / `hits_upper_bound` (:137)


def secondary_cracker(backwords, words, config, func_threshold: Tuple[int, int], **kwargs):
    save_dir, tag, splitter = kwargs['save'], kwargs['tag'], kwargs['splitter']
    counts, words = backwords_counter(
        nwords_list=kwargs['training'], splitter=splitter, start_chr=config['start_chr'], end_chr=config['end_chr'],
        start4words=kwargs['start4words'], step4words=kwargs['skip4words'], max_gram=kwargs['max_gram'],
        nwords_dict=backwords, words=words, threshold=kwargs['threshold'])
    config['training_list'].append(f"{kwargs['sign']}")
    with open(os.path.join(save_dir, f"model-to-crack-{tag}.pickle"), 'wb') as fd:
        pickle.dump((counts, words, config), file=fd)
    model = BackWordsSecondaryMonteCarlo((counts, words, config), max_iter=kwargs['max_iter'])

    by_samples = kwargs['using_sample_attack']
    sampled = {} if by_samples else None
    keys = model.sample(size=kwargs['size'], sampled_pwds=sampled)
    if by_samples:
        # samples in ascending -log2 p (ties: first sampled first); a string's guess number is
        # 1 + the number of samples, with multiplicity, written before it (:45-51)
        with open(os.path.join(save_dir, f"samples-{tag}.txt"), 'w') as fout:
            before = 1
            for pwd, (prob, cnt) in sorted(sampled.items(), key=lambda kv: kv[1][0]):
                fout.write(f"{pwd}\t{prob:.8f}\t{cnt}\n")
                sampled[pwd] = before
                before += cnt
    mc = MonteCarloLib(keys)
    gc = mc.ml2p_iter2gc(minus_log_prob_iter=model.parse_file(kwargs['testing'], using_component=True))

    next_training: List[str] = []
    done, cum = kwargs['already_cracked'], kwargs['cum']
    gn_bound, hits_bound = func_threshold
    hits, max_gn = 0, 0
    with open(os.path.join(save_dir, f"cracked-{tag}.txt"), 'w') as fout:
        for parts, prob, num, gn, _cracked, _ratio in gc:
            pwd = splitter.join(parts)
            if pwd in done:
                continue
            hit = by_samples and pwd in sampled
            if hit:
                gn = sampled[pwd]
            if hit or (not by_samples and gn < gn_bound and hits < hits_bound):
                hits += 1
                max_gn = max(max_gn, gn)
                next_training.extend([pwd] * num)
                cum.append((pwd, prob, num, gn))
                fout.write(f"{pwd}\t{prob:.8f}\t{num}\t{gn}\n")
                done.add(pwd)
    keep = kwargs['secondary_sample']
    if keep < len(next_training):
        print(f"We sample {keep} passwords to perform secondary training in the next round", file=sys.stderr)
        next_training = random.sample(next_training, keep)
        with open(os.path.join(save_dir, f"sampled-{tag}.txt"), 'w') as fout:
            fout.writelines(f"{pwd}\n" for pwd in next_training)
    return counts, words, config, next_training, max_gn


# the reference's command line (backwords_secondary_main.py:92-133): flag, dest, type, default, help
_FILE_FLAGS = (
    ("-i", "--training", "training", argparse.FileType('r'), "training passwords, one per line"),
    ("-t", "--testing", "testing", argparse.FileType('r'), "test passwords, one per line"),
    ("-s", "--save", "save", str, "folder that receives every result file"),
)
_OPTIONS = (
    ("--size", "size", int, 100000, "Monte Carlo samples per round"),
    ("--secondary-sample", "secondary_sample", int, 10000000000,
     "upper bound on the cracked passwords that feed the next round (drawn with random.sample beyond it)"),
    ("--splitter", "splitter", str, "empty", "column separator of the input: empty | space | tab | a literal"),
    ("--start4word", "start4words", int, 0, "column of the first word"),
    ("--skip4word", "skip4words", int, 1, "step from one word column to the next (tags sit between them)"),
    ("--max-gram", "max_gram", int, 256, "longest context"),
    ("--threshold", "threshold", int, 10, "counts below it are pruned"),
    ("--max-iter", "max_iter", int, 10 ** 20, "bound on the segmentations tried by calc_ml2p"),
)


def build_cli() -> argparse.ArgumentParser:
    cli = argparse.ArgumentParser('Backwords secondary main')
    for short, flag, dest, kind, text in _FILE_FLAGS:
        cli.add_argument(short, flag, dest=dest, type=kind, required=True, help=text)
    cli.add_argument("--strategy", dest="strategy", required=True, type=str, nargs="+",
                     help="guesses <g1> <g2> ... | hits <h1> <h2> ... | auto_hits <factor> <base> <termination> | "
                          "samples <rounds> (the sampled strings themselves are the guesses)")
    for flag, dest, kind, default, text in _OPTIONS:
        cli.add_argument(flag, dest=dest, type=kind, required=False, default=default, help=text)
    return cli


def plan_rounds(strategy: List[str], size: int):
    """(per-round (guess bound, hit bound), per-round sign, sample attack?) -- reference :136-170."""
    kind, values = strategy[0], strategy[1:]
    if kind == 'guesses':
        return [(int(v), GN_CAP) for v in values], [f"guesses-{int(v):,}" for v in values], False
    if kind == 'hits':
        return [(GN_CAP, int(v)) for v in values], [f"hits-{int(v):,}" for v in values], False
    if kind == 'auto_hits':
        factor, base, termination = int(values[0]), int(values[1]), int(values[2])
        end = math.ceil(math.log(termination / max(base, 1)) / math.log(max(factor, 1)))
        steps = list(range(1, end))
        return [(GN_CAP, factor ** v * base) for v in steps], [f"auto_hits-{v:,}" for v in steps], False
    rounds = int(values[0])
    return [(GN_CAP, GN_CAP)] * rounds, [f"samples-{size}"] * rounds, True


def join_ranks() -> int:
    """Under torch.distributed.run (WORLD_SIZE > 1): one rank per GPU over RCCL -- or, when the box has
    fewer GPUs than ranks (a rehearsal), shared GPUs over gloo -- and sharding switched on.  Returns
    this process's rank (0 when not distributed)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0
    import torch
    import torch.distributed as dist
    from . import dist as mcd
    if not dist.is_initialized():
        local, n_dev = int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count()
        if n_dev >= int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.cuda.set_device(local % max(n_dev, 1))
            # (ranks share a GPU: fine, the library's sort makes progress under any residency)
            dist.init_process_group("gloo")
    mcd.use_group(True)
    return dist.get_rank()


def wrapper(argv=None):
    args = build_cli().parse_args(argv)
    rank = join_ranks()
    if rank != 0:
        import tempfile
        args.save = tempfile.mkdtemp(prefix=f"mc-rank{rank}-")       # only rank 0's files are the result
    permits = {'guesses', 'hits', 'samples', 'auto_hits'}
    if args.strategy[0] not in permits:
        print(f"strategy should be one of `{', '.join(permits)}`", file=sys.stderr)
        return
    if len(args.strategy) < 2:
        print("strategy should have at least 2 values", file=sys.stderr)
        return
    print(f"using {args.strategy[0]}", file=sys.stderr)
    bounds, signs, by_samples = plan_rounds(args.strategy, args.size)
    args.splitter = {'empty': '', 'space': ' ', 'tab': '\t'}.get(args.splitter.lower(), args.splitter)
    start_chr, end_chr = '\x03', '\x00'                                 # reversed w.r.t. the plain simulators (:175)
    config = {'start_chr': start_chr, 'end_chr': end_chr, 'max_gram': args.max_gram, 'threshold': args.threshold,
              'training_list': [args.training.name]}
    if not os.path.exists(args.save):
        os.mkdir(args.save)
    counts, words, training = None, None, args.training
    already_cracked = set()
    print(f"We will have {len(bounds)} rounds", file=sys.stderr, end=', ')
    cums: List[List[Tuple[str, float, int, int]]] = []
    max_gns: List[int] = []
    for idx, bound in enumerate(bounds):
        print(f"The {idx}-th iteration", file=sys.stderr)
        cum: List[Tuple[str, float, int, int]] = []
        counts, words, config, training, max_gn = secondary_cracker(
            counts, words, config=config, func_threshold=bound, training=training, splitter=args.splitter,
            start4words=args.start4words, skip4words=args.skip4words, max_gram=args.max_gram, size=args.size,
            max_iter=args.max_iter, testing=args.testing, save=args.save, secondary_sample=args.secondary_sample,
            already_cracked=already_cracked, cum=cum, threshold=args.threshold, sign=signs[idx],
            using_sample_attack=by_samples, tag=f"iter-{idx}")
        cums.append(cum)
        max_gns.append(max_gn)
        if max_gn >= GN_CAP:
            print(f"Too large guess number reached: {max_gn}, the training process is terminated", file=sys.stderr)
            break
    # the last round's cracked passwords go into the final model
    counts, words = backwords_counter(
        training, splitter=args.splitter, start_chr=start_chr, end_chr=end_chr, start4words=args.start4words,
        step4words=args.skip4words, max_gram=args.max_gram, nwords_dict=counts, words=words, threshold=args.threshold)
    with open(os.path.join(args.save, "final_model.pickle"), 'wb') as fout:
        pickle.dump((counts, words, config), file=fout)
    print("Training phase done.", file=sys.stderr)
    model = BackWordsSecondaryMonteCarlo((counts, words, config), max_iter=args.max_iter)
    mc = MonteCarloLib(model.sample(size=args.size))
    gc = mc.ml2p_iter2gc(minus_log_prob_iter=model.parse_file(args.testing))
    # what the final model alone cracks
    with open(os.path.join(args.save, "iter_result.txt"), 'w') as fout:
        cum = []
        for pwd, prob, num, gn, cracked, ratio in gc:
            fout.write(f"{pwd}\t{prob:.8f}\t{num}\t{gn}\t{cracked}\t{ratio:5.2f}\n")
            if pwd not in already_cracked:
                cum.append((pwd, prob, num, gn))
        cums.append(cum)
    # the union of all rounds: a round's guesses come after every guess of the rounds before it
    with open(os.path.join(args.save, "sectional_result.txt"), "w") as fout:
        cracked, total = 0, wc_l(args.testing)
        for offset, cum in zip([0, *max_gns], cums):
            for pwd, prob, num, gn in cum:
                cracked += num
                fout.write(f"{pwd}\t{prob:.8f}\t{num}\t{gn + offset}\t{cracked}\t{cracked / total * 100:5.2f}\n")
    with open(os.path.join(args.save, "config.json"), 'w') as fout:
        json.dump(config, fp=fout, indent=2)
    args.testing.close()


if __name__ == '__main__':
    wrapper()
