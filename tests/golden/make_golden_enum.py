#!/usr/bin/env python3
"""Golden vectors for the threshold enumerators: runs the REFERENCE's own
nwords/nwords_enumerator.py and backwords/backwords_enumerator.py (with the reference's trainers) on the
committed corpora and records what they write.  Build container only (imports /root/reference)."""
import hashlib
import io
import json
import os
import sys

sys.dont_write_bytecode = True
os.environ["TQDM_DISABLE"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

from backwords.backwords_enumerator import enumerator as back_enum  # noqa: E402
from backwords.backwords_trainer import backwords_counter  # noqa: E402
from nwords.nwords_enumerator import enumerator as nw_enum  # noqa: E402
from nwords.nwords_trainer import nwords_counter  # noqa: E402


def corpus(name):
    return open(os.path.join(HERE, name), encoding="utf-8")


def record(text: str, meta: dict):
    lines = text.split("\n")[:-1]
    out = dict(meta, n_lines=len(lines), sha256=hashlib.sha256(text.encode("utf-8", "surrogatepass")).hexdigest(),
               early_end_lines=sum(1 for ln in lines if meta["end_chr"] in ln.split("\t")[0]))
    if len(text) <= 300_000:
        out["lines"] = lines
    else:
        out["head"], out["tail"] = lines[:200], lines[-200:]
    return out


cases = {}
# fixed order, single characters (nwords_enum.py:31-36 passes end '\x00', start '\x03')
for name, n, splitter, train, thr, min_len in (("nwords_n3_char", 3, "", "train_char.txt", 17.0, 4),
                                               ("nwords_n4_char_min6", 4, "", "train_char.txt", 14.5, 6),
                                               ("nwords_n2_tab", 2, "\t", "train_tab.txt", 15.0, 4)):
    table, _ = nwords_counter(corpus(train), n=n, splitter=splitter, end_chr="\x00", start_chr="\x03",
                              start4words=0, skip4words=1)
    buf = io.StringIO()
    nw_enum(table, threshold=thr, end_chr="\x00", start_chr="\x03", min_len=min_len, f_save=buf, order=n - 1)
    cases[name] = record(buf.getvalue(), {"kind": "nwords", "n": n, "splitter": splitter, "train": train,
                                          "threshold": thr, "min_len": min_len, "start_chr": "\x03", "end_chr": "\x00"})
# backoff (backwords_enum.py:19-20: backwords_counter(file, '', '\x00', '\x03', 0, 1, 10, 256))
for name, splitter, train, count_thr, thr, min_len in (("backwords_thr10_char", "", "train_char.txt", 10, 16.0, 4),
                                                       ("backwords_thr3_char_min7", "", "train_char.txt", 3, 13.0, 7),
                                                       ("backwords_thr3_tab", "\t", "train_tab.txt", 3, 14.0, 4)):
    table, _ = backwords_counter(corpus(train), splitter, "\x00", "\x03", 0, 1, count_thr, 256)
    buf = io.StringIO()
    back_enum(table, thr, "\x00", "\x03", min_len, buf)
    cases[name] = record(buf.getvalue(), {"kind": "backwords", "splitter": splitter, "train": train,
                                          "count_threshold": count_thr, "threshold": thr, "min_len": min_len,
                                          "start_chr": "\x00", "end_chr": "\x03"})
for k, v in cases.items():
    print(k, v["n_lines"], "lines,", v["early_end_lines"], "with an early end symbol inside")
with open(os.path.join(HERE, "enumerators.json"), "w") as f:
    json.dump(cases, f, ensure_ascii=True, indent=0)
