"""Incremental (secondary) backoff trainer: same contract as the reference's
backwords/backwords_secondary_trainer.py:23-121 -- raw counts are merged across
rounds (`backwords_counter`), probabilities are derived afterwards (`freq2prob`)."""
import sys
from typing import Dict, List, Tuple, Union

from .._counting import float_sum, read_lines, row_total, split_line, transitions
from .backwords_trainer import group_by_length


# plain-dict models (pickles) are taken over by the native trainer when they are well-formed; tests
# switch this off to keep the dict-based restatement of the reference beside it
IMPORT_PLAIN_COUNTS = True


def parse_line(line: str, splitter: str, start4words: int, step4words: int):
    return split_line(line, splitter, start4words, step4words)


def backwords_counter(nwords_list, splitter: str, start_chr: str, end_chr: str,
                      start4words: int, step4words: int, max_gram: int, threshold: int,
                      nwords_dict: Dict[Tuple, Dict[str, int]] = None,
                      words: Dict[str, int] = None):
    from ..trained import NativeCounts, one_code_point, splitter_code_point
    sig = (splitter_code_point(splitter), one_code_point(start_chr), one_code_point(end_chr))
    given_counts = nwords_dict if type(nwords_dict) is dict else None      # the caller's own objects: the
    given_words = words if type(words) is dict else None                    # reference updates them in place (:44-78)
    if IMPORT_PLAIN_COUNTS and None not in sig and isinstance(nwords_dict, dict) and nwords_dict and words:
        # a model that comes as plain dicts (a pickle of an earlier run, the reference's or ours) resumes on
        # the native path when it has the shape the secondary trainer builds
        native = NativeCounts.from_plain(nwords_dict, words, start_chr, end_chr)
        if native is not None:
            nwords_dict = native
    if None not in sig and (nwords_dict is None or (isinstance(nwords_dict, NativeCounts)
                                                    and nwords_dict.signature in (None, sig))):
        # native path (csrc/mc_train.cpp): the counts live in C++ and are merged there
        if isinstance(nwords_list, list):
            if len(nwords_list) == 0:
                print("No passwords for training, early return!", file=sys.stderr)
                return ({} if nwords_dict is None else nwords_dict), ({} if words is None else words)
            text = "".join(ln if ln.endswith("\n") else ln + "\n" for ln in nwords_list)
        else:
            text = read_lines_keep_open(nwords_list).read()
        counts = NativeCounts() if nwords_dict is None else nwords_dict
        counts.merge(text, sig[0], start4words, step4words, max_gram, threshold, sig[1], sig[2])
        merged_words = counts.words()
        if given_words is not None:
            given_words.clear()
            given_words.update(merged_words)
        if given_counts is not None:
            # a caller that keeps using the dicts it passed in (the reference's main pickles its own `words`
            # argument, not the return value) sees the merge, as with the reference; this happens once per
            # resumed pickle -- later rounds pass the NativeCounts this call returns
            given_counts.clear()
            given_counts.update({ctx: dict(row) for ctx, row in counts.items()})
        return counts, (given_words if given_words is not None else merged_words)
    if isinstance(nwords_dict, NativeCounts):
        nwords_dict = {ctx: dict(row) for ctx, row in nwords_dict.items()}
    if nwords_dict is None:
        nwords_dict, words = {}, {}
    if isinstance(nwords_list, list) and len(nwords_list) == 0:
        print("No passwords for training, early return!", file=sys.stderr)
        return nwords_dict, words
    uniq: Dict[Tuple, int] = {}
    longest = 2
    for line in (nwords_list if isinstance(nwords_list, list) else read_lines_keep_open(nwords_list)):
        secs = tuple([start_chr] + split_line(line, splitter, start4words, step4words) + [end_chr])
        for s in secs:
            words[s] = words.get(s, 0) + 1
            if s != start_chr:
                zero = nwords_dict.setdefault((), {})
                zero[s] = zero.get(s, 0) + 1
        uniq[secs] = uniq.get(secs, 0) + 1
        longest = max(longest, len(secs))
    groups = group_by_length(uniq)
    for n in range(2, min(max_gram, longest) + 1):
        fresh = {}
        for ln, grp in groups.items():
            if ln >= n:
                transitions(grp, n - 1, into=fresh)
        if not fresh:
            break
        # a context unseen so far is admitted only with a transition at/above the threshold
        for ctx, row in fresh.items():
            have = nwords_dict.get(ctx)
            if have is None:
                if any(c >= threshold for c in row.values()):
                    nwords_dict[ctx] = row
                continue
            for s, c in row.items():
                have[s] = have.get(s, 0) + c
    return nwords_dict, words


def read_lines_keep_open(fd):
    """The secondary trainer, unlike the primary ones, leaves its file open and
    rewound-on-entry (the reference's wc_l does the seek, FileLib.py:13-14)."""
    if hasattr(fd, "seekable") and fd.seekable():
        fd.seek(0)
    return fd


def freq2prob(nwords_dict: Dict[Tuple, Dict[str, int]], threshold: int) -> Dict:
    table: Dict[Tuple, Dict[str, float]] = {}
    for ctx, row in sorted(nwords_dict.items(), key=lambda kv: len(kv[0])):   # parents before children
        total = row_total(row)
        kept = {s: c / total for s, c in row.items() if c >= threshold}
        if not kept:
            continue
        if len(kept) < len(row) and ctx:
            missing = 1.0 - float_sum(kept.values())
            for s, p in table[ctx[1:]].items():
                kept[s] = kept.get(s, .0) + p * missing
        table[ctx] = kept
    return table
