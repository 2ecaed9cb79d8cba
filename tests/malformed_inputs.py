"""Malformed and extreme inputs through the host-side C ABI (table compiler, trainers, ingest, writers): run as a
script by tests/test_sanitized_host.py inside a process whose host entry points come from the library built under
AddressSanitizer + UBSan (MC_HOST_SANITIZED), and imported by the same test file for a plain run.  Every call must
return a status (never crash, never read outside its input); the statuses and the results that the reference's
behaviour fixes are asserted.

Reference error paths these stand in for: a bad model path / a `max_gram == 1` exit (bpeX/modelreader.py:8-12,
backwords/backwords_trainer.py:49-51), "error1" of the v2 trainer (nwords_v2/nwords_trainer_v2.py:32-33);
everything else here is input the reference would hit a Python exception on (UnicodeDecodeError on invalid
UTF-8 in text mode) and the C ABI answers with MC_ERR_ARG."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from montecarlo_amd import _native as nv

u32, u64 = np.uint32, np.uint64


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def ingest(data: bytes):
    h = C.c_void_p()
    rc = nv.lib.mc_lines_ingest(data, len(data), C.byref(h))
    if rc != 0:
        return rc, None
    info = np.zeros(4, dtype=u64)
    nv.lib.mc_lines_array(h, nv.LINES_INFO, vp(info), info.nbytes)
    cnt = np.zeros(int(info[0]), dtype=u64)
    nv.lib.mc_lines_array(h, nv.LINES_COUNT, vp(cnt), cnt.nbytes)
    off = np.zeros(int(info[0]) + 1, dtype=u64)
    nv.lib.mc_lines_array(h, nv.LINES_OFF, vp(off), off.nbytes)
    cps = np.zeros(int(info[3]), dtype=u32)
    nv.lib.mc_lines_array(h, nv.LINES_CPS, vp(cps), cps.nbytes)
    nv.lib.mc_lines_free(h)
    lines = ["".join(map(chr, cps[int(off[i]):int(off[i + 1])])) for i in range(len(cnt))]
    return 0, dict(zip(lines, cnt.tolist()))


def train(kind: str, data: bytes, **kw):
    h = C.c_void_p()
    if kind == "nwords":
        rc = nv.lib.mc_train_nwords(data, len(data), kw.get("n", 3), kw.get("splitter", 0), 0, 1, 0, 3, C.byref(h))
    elif kind == "backwords":
        rc = nv.lib.mc_train_backwords(data, len(data), kw.get("splitter", 0), 0, 1, kw.get("threshold", 2),
                                       kw.get("max_gram", 256), 0, 3, C.byref(h))
    else:
        rc = nv.lib.mc_train_v2(data, len(data), kw.get("n", 4), 3, kw.get("threshold", 2), C.byref(h))
    rows = None
    if rc == 0:
        size = nv.lib.mc_trained_array(h, nv.TRAINED_CTX_OFF, None, 0)
        rows = size // 4 - 1
        t = C.c_void_p()
        rc2 = nv.lib.mc_trained_compile(h, nv.KIND_V2 if kind == "v2" else (nv.KIND_NGRAM if kind == "nwords" else nv.KIND_BACKOFF),
                                        kw.get("n", 3) - 1, 4, 256, 0, 0, C.byref(t))
        if rc2 == 0:
            nv.lib.mc_table_free(t)
        nv.lib.mc_trained_free(h)
    return rc, rows


def main():
    ok_corpus = ("\n".join(["password", "passw0rd", "letmein", "dragon", "pass", "password"] * 5) + "\n").encode()
    # ---- ingest ----
    assert ingest(b"") == (0, {})                                              # an empty file
    assert ingest(b"\n\n\n") == (0, {"": 3})
    assert ingest(b"abc") == (0, {"abc": 1})                                   # no newline at the end
    assert ingest(b"a\r\nb\r\na\r\n") == (0, {"a": 2, "b": 1})                 # \r\n: stripped
    rc, got = ingest(b"a\rb\rc\n")                                             # \r inside a line stays (one line)
    assert rc == 0 and got == {"a\rb\rc": 1}
    rc, got = ingest(b"p\x00w\nq\n")                                           # NUL is a character like any other
    assert rc == 0 and got == {"p\x00w": 1, "q": 1}
    long_line = b"x" * (1 << 20)
    rc, got = ingest(long_line + b"\n" + long_line + b"\nshort\n")             # 1 MB lines
    assert rc == 0 and got == {"x" * (1 << 20): 2, "short": 1}
    for bad in (b"ab\xff\n", b"\xc3\n", b"\xe2\x82\n", b"\xf0\x9f\x98\n", b"\xc0\xaf\n", b"\xe0\x80\xaf\n",
                b"\xf4\x90\x80\x80\n", b"ok\n\x80"):                           # invalid UTF-8 of every kind
        rc, _ = ingest(bad)
        assert rc == -1, (bad, rc)
    assert ingest("päss\n日本語\n😀\n".encode()) == (0, {"päss": 1, "日本語": 1, "😀": 1})
    assert ingest("a\udc80b\n".encode("utf-8", "surrogatepass")) == (0, {"a\udc80b": 1})   # Python's carrier of a raw byte
    assert nv.lib.mc_lines_ingest(None, 5, C.byref(C.c_void_p())) == -1
    # ---- trainers ----
    for kind in ("nwords", "backwords"):
        assert train(kind, ok_corpus)[0] == 0
        assert train(kind, b"")[0] in (0, -1, -2)                              # no lines: a status, not a crash
        assert train(kind, b"\n\n")[0] in (0, -1, -2)
        assert train(kind, b"ab\xffcd\n")[0] == -1                             # invalid UTF-8
        assert train(kind, b"p\x00w\np\x00w\n" * 3)[0] in (0, -1, -2)          # NUL = the start symbol's code point here
        assert train(kind, long_line + b"\n")[0] in (0, -1, -2)                # one 1 MB line
        assert train(kind, ok_corpus.replace(b"\n", b"\r"))[0] in (0, -1, -2)  # \r-only line ends: one long line
        assert train(kind, ok_corpus, splitter=ord("\t"))[0] in (0, -1, -2)    # a splitter no line contains
    assert train("backwords", ok_corpus, max_gram=1)[0] == 0                   # (max(2, max_gram), backwords_trainer.py:48)
    assert train("backwords", ok_corpus, threshold=10 ** 6)[0] in (0, -1, -2)  # nothing reaches the threshold (:49-51)
    assert train("nwords", ok_corpus, n=0)[0] == -1
    v2 = b"password\tpass\tL\tword\tL\n" * 4
    assert train("v2", v2, n=4)[0] == 0
    assert train("v2", b"password\tpass\tL\n", n=4)[0] == -1                    # segments do not rebuild the password ("error1")
    assert train("v2", b"ab\tab\tL\n", n=4)[0] == -1                            # a password shorter than the trainer takes ("error1")
    assert train("v2", b"password\n", n=4)[0] in (0, -1)                       # no segments at all
    assert train("v2", b"\xffpassword\tpass\tL\n", n=4)[0] == -1
    # ---- writers ----
    n = 3
    mlp = np.array([1.5, 2.25, 1e17], dtype=np.float64)
    ints = np.arange(1, n + 1, dtype=np.int64)
    ratio = np.array([0.5, 50.0, 100.0])
    dst = np.zeros(1000, dtype=np.uint8)
    text = "a\nbb\nccc".encode()
    size = nv.lib.mc_format_gc(text, len(text), n, vp(mlp), vp(ints), vp(ints), vp(ints), vp(ratio), vp(dst), dst.size)
    want = "".join(f"{p}\t{m:.8f}\t{a}\t{a}\t{a}\t{r:5.2f}\n" for p, m, a, r in zip(("a", "bb", "ccc"), mlp, ints, ratio))
    assert size > 0 and bytes(dst[:size]).decode() == want
    assert nv.lib.mc_format_gc(text, len(text), n, vp(mlp), vp(ints), vp(ints), vp(ints), vp(ratio), vp(dst), 10) < 0   # too small
    assert nv.lib.mc_format_gc(text, len(text), 5, vp(mlp), vp(ints), vp(ints), vp(ints), vp(ratio), vp(dst), dst.size) < 0  # fewer lines than rows
    cps = np.array([0x61, 0xD800, 0x62], dtype=u32)                            # a lone surrogate has no UTF-8 form
    off = np.array([0, 1, 2, 3], dtype=u64)
    order = np.array([2, 1, 0], dtype=u32)
    assert nv.lib.mc_format_gc_cps(vp(cps), vp(off), 3, vp(order), n, vp(mlp), vp(ints), vp(ints), vp(ints), vp(ratio),
                                   vp(dst), dst.size) == -1
    cps[1] = 0x1F600
    size = nv.lib.mc_format_gc_cps(vp(cps), vp(off), 3, vp(order), n, vp(mlp), vp(ints), vp(ints), vp(ints), vp(ratio),
                                   vp(dst), dst.size)
    assert size > 0 and bytes(dst[:size]).decode().split("\n")[0].startswith("b\t")
    order[0] = 7                                                               # a row that names a line beyond the table
    assert nv.lib.mc_format_gc_cps(vp(cps), vp(off), 3, vp(order), n, vp(mlp), vp(ints), vp(ints), vp(ints), vp(ratio),
                                   vp(dst), dst.size) < 0
    # ---- counts import (the (counts, words, config) pickle of backwords_secondary_main.py:28-32 as arrays) ----
    def imp(ent_sym, word_sym=(0, 1, 2), ctx_units=(0,), n_sym=3):
        k = C.c_void_p()
        assert nv.lib.mc_counts_new(C.byref(k)) == 0
        sym_text_off = np.arange(n_sym + 1, dtype=u32)
        sym_text = np.array([ord("a") + i for i in range(n_sym)], dtype=u32)
        word_sym = np.array(word_sym, dtype=u32)
        word_cnt = np.full(len(word_sym), 5, dtype=u64)
        ctx_off = np.array([0, len(ctx_units)], dtype=u32)
        ctx_units = np.array(ctx_units, dtype=u32)
        row_off = np.array([0, len(ent_sym)], dtype=u64)
        ent_sym = np.array(ent_sym, dtype=u32)
        ent_cnt = np.full(len(ent_sym), 3, dtype=u64)
        rc = nv.lib.mc_counts_import(k, vp(sym_text_off), vp(sym_text), n_sym, vp(word_sym), vp(word_cnt), len(word_sym),
                                     vp(ctx_off), vp(ctx_units), vp(row_off), vp(ent_sym), vp(ent_cnt), 1, 3, 0)
        if rc == 0:
            t = C.c_void_p()
            rc2 = nv.lib.mc_counts_train(k, 1, C.byref(t))
            if rc2 == 0:
                nv.lib.mc_trained_free(t)
        nv.lib.mc_counts_free(k)
        return rc
    assert imp([0, 1]) == 0
    assert imp([0, 9]) == -1                                                   # a row names a symbol the pickle does not define
    assert imp([0, 1], word_sym=(0, 1, 7)) == -1                               # so does `words`
    assert imp([0, 1], ctx_units=(5,)) == -1                                   # so does a context
    assert imp([0, 0]) in (0, -1)                                              # the same symbol twice in a row
    print("malformed_inputs ok:", len(getattr(nv.lib, "sanitized_host_entry_points", [])), "entry points from the sanitized library")


if __name__ == "__main__":
    sys.exit(main())
