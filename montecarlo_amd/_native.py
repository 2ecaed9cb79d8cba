"""ctypes binding of libmi355mc.so (include/mi355mc.h).

There is no fallback: if the library is missing the import fails loudly, so a
product path can never silently run on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  It must be in the process BEFORE
# libmi355mc.so is loaded, so that the loader resolves our DT_NEEDED libamdhip64.so.7 to that same
# copy: two HIP runtimes in one process leave the second without a device.
import torch  # noqa: F401  (device memory, streams and torch.distributed are torch's anyway)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MC_LIB_PATH: another build of the same sources, for A/B timing of a compile-time choice; development only)
LIB_PATH = os.environ.get("MC_LIB_PATH") or os.path.join(_HERE, "libmi355mc.so")

MC_NONE = 0xFFFFFFFF
KIND_NGRAM, KIND_BACKOFF, KIND_V2, KIND_BPE = 1, 2, 3, 4
FLAG_KEEP_LEN_ON_RESTART = 1
FLAG_NO_DENSE = 2
(ARR_STATE, ARR_DRAW, ARR_CUM, ARR_PAYLOAD, ARR_TRANS, ARR_CTXHASH, ARR_SYMCHARS, ARR_SPAY, ARR_RCUM, ARR_WORDHASH,
 ARR_SYMTEXTOFF, ARR_SYMTEXT, ARR_DENSE, ARR_WORDCK) = range(14)
(INFO_NCTX, INFO_NENT, INFO_NSYM, INFO_START_STATE, INFO_ROOT, INFO_TRANS_CAP, INFO_CTX_CAP, INFO_GUIDE_LEN,
 INFO_DEVICE_BYTES, INFO_MAX_ROW, INFO_KIND, INFO_MULTICHAR, INFO_START_DRAW, INFO_DIRECT_BUCKETS,
 INFO_ONE_BUCKETS, INFO_MULTI_BUCKETS) = range(16)
DRAW_DIRECT, DRAW_ONE, DRAW_MULTI = 0, 1, 2
(TRAINED_CTX_OFF, TRAINED_CTX_UNITS, TRAINED_ROW_OFF, TRAINED_ENT_SYM, TRAINED_ENT_PROB, TRAINED_ENT_CUM,
 TRAINED_ENT_ML2P, TRAINED_SYM_TEXT_OFF, TRAINED_SYM_TEXT, TRAINED_WORD_SYM, TRAINED_WORD_COUNT) = range(11)
LINES_COUNT, LINES_OFF, LINES_CPS, LINES_INFO = range(4)
ENUM_ML2P, ENUM_LEN, ENUM_SYMS = range(3)
COUNTS_ENT_COUNT = 16
ERR_DIVERGED = -5
ERR_UNSUITED = -7

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


class ModelDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32), ("order", C.c_uint32), ("n_ctx", C.c_uint32), ("n_sym", C.c_uint32),
        ("n_ent", C.c_uint64),
        ("ctx_off", u32p), ("ctx_units", u32p), ("row_off", u64p), ("ent_sym", u32p),
        ("ent_cum", f64p), ("ent_ml2p", f64p),
        ("sym_unit_off", u32p), ("sym_units", u32p), ("sym_chars", u32p),
        ("start_units", u32p), ("n_start", C.c_uint32),
        ("end_sym", C.c_uint32), ("min_len", C.c_uint32), ("max_len", C.c_uint32), ("flags", C.c_uint32),
        ("guide_factor", C.c_uint32),
        ("sym_text_off", u32p), ("sym_text", u32p),
        ("seg_off", u32p), ("seg_row", u32p), ("seg_len", u32p),
        ("n_labels", C.c_uint32), ("n_np", C.c_uint32),
        ("lab_off", u32p), ("lab_units", u32p), ("lab_cand_off", u32p), ("lab_cand", u32p),
        ("np_len", u32p), ("np_off", u32p), ("np_cand", u32p),
    ]


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmi355mc error {code}: {msg}")
        self.code = code


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m montecarlo_amd.build` (hipcc, gfx950). "
            "montecarlo_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    sig = {
        "mc_abi_version": (C.c_int, []),
        "mc_last_error": (C.c_char_p, []),
        "mc_table_compile": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(vp)]),
        "mc_table_free": (None, [vp]),
        "mc_table_array": (C.c_int64, [vp, C.c_int, vp, C.c_size_t]),
        "mc_table_info": (C.c_int, [vp, u64p]),
        "mc_ctx_hash": (C.c_uint64, [u32p, C.c_uint32]),
        "mc_text_hash": (C.c_uint64, [u32p, C.c_uint32]),
        "mc_train_nwords": (C.c_int, [C.c_char_p, C.c_size_t] + [C.c_uint32] * 6 + [C.POINTER(vp)]),
        "mc_train_backwords": (C.c_int, [C.c_char_p, C.c_size_t] + [C.c_uint32] * 7 + [C.POINTER(vp)]),
        "mc_counts_new": (C.c_int, [C.POINTER(vp)]),
        "mc_counts_free": (None, [vp]),
        "mc_counts_merge": (C.c_int, [vp, C.c_char_p, C.c_size_t] + [C.c_uint32] * 7),
        "mc_counts_train": (C.c_int, [vp, C.c_uint32, C.POINTER(vp)]),
        "mc_counts_import": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, C.c_uint32,
                                       C.c_uint32, C.c_uint32]),
        "mc_counts_array": (C.c_int64, [vp, C.c_int, vp, C.c_size_t]),
        "mc_train_v2": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
        "mc_trained_free": (None, [vp]),
        "mc_trained_array": (C.c_int64, [vp, C.c_int, vp, C.c_size_t]),
        "mc_trained_compile": (C.c_int, [vp] + [C.c_uint32] * 6 + [C.POINTER(vp)]),
        "mc_enumerate": (C.c_int, [vp, C.c_double, C.c_uint32, C.c_uint32, C.c_uint64, vp, C.POINTER(vp)]),
        "mc_enum_info": (C.c_int, [vp, u64p]),
        "mc_enum_fetch": (C.c_int64, [vp, C.c_int, vp, C.c_size_t]),
        "mc_enum_free": (None, [vp]),
        "mc_format_gc": (C.c_int64, [C.c_char_p, C.c_size_t, C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_size_t]),
        "mc_format_gc_cps": (C.c_int64, [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_size_t]),
        "mc_lines_ingest": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(vp)]),
        "mc_lines_free": (None, [vp]),
        "mc_lines_array": (C.c_int64, [vp, C.c_int, vp, C.c_size_t]),
        "mc_model_upload": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
        "mc_model_destroy": (None, [vp]),
        "mc_model_check": (C.c_int, [vp, vp, u64p]),
        "mc_sample": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp, vp, vp, C.c_uint32, vp]),
        "mc_sample_tune": (C.c_int, [vp, C.c_int, C.c_float, C.c_int]),
        "mc_sample_carry": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp, vp, vp, C.c_uint32,
                                      vp, vp, C.c_uint64, vp]),
        "mc_sample_injected": (C.c_int, [vp, C.c_uint64, vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, vp]),
        "mc_sample_dedup": (C.c_int, [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, vp, vp, vp, vp, vp]),
        "mc_score": (C.c_int, [vp, vp, vp, C.c_uint64, vp, vp]),
        "mc_score_words_workspace_bytes": (C.c_size_t, [vp, C.c_uint64, C.c_uint32]),
        "mc_score_words": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint64, vp, vp, C.c_uint32, vp,
                                     C.c_size_t, vp]),
        "mc_score_bpe": (C.c_int, [vp, vp, vp, vp, C.c_uint64, vp, vp]),
        "mc_rank_workspace_bytes": (C.c_size_t, [C.c_uint64]),
        "mc_rank_build": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_size_t, vp]),
        "mc_rank_check": (C.c_int, [vp, C.c_uint64, vp, u64p]),
        "mc_rank_debug_fail": (None, [C.c_int]),
        "mc_rank_debug_search": (None, [C.c_int]),
        "mc_workspace_forget": (C.c_int, [vp, C.c_uint64, vp]),
        "mc_rank_noalloc_max_keys": (C.c_uint64, []),
        "mc_rank_build_noalloc": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, vp, vp]),
        "mc_order_workspace_bytes": (C.c_size_t, [C.c_uint64]),
        "mc_order_keys": (C.c_int, [vp, C.c_uint64, vp, vp, vp, C.c_size_t, vp]),
        "mc_order_keys_begin": (C.c_int, [vp, C.c_uint64, vp, vp, vp, C.c_size_t, vp]),
        "mc_order_keys_end": (C.c_int, [vp, C.c_uint64, vp, vp, vp, C.c_size_t, vp]),
        "mc_guess_numbers_workspace_bytes": (C.c_size_t, [C.c_uint64]),
        "mc_guess_numbers": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp, vp, vp, C.c_size_t, vp]),
        "mc_comm_unique_id": (C.c_int, [vp]),
        "mc_comm_init": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
        "mc_allreduce_rank": (C.c_int, [vp, vp, vp, C.c_uint64, vp]),
        "mc_comm_destroy": (None, [vp]),
        "mc_rank_query": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp]),
        "mc_rank_runs_workspace_bytes": (C.c_size_t, [C.c_uint64]),
        "mc_rank_runs_capacity": (C.c_uint64, [C.c_uint64]),
        "mc_rank_runs_max_keys": (C.c_uint64, []),
        "mc_rank_runs_build": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, vp, vp, C.c_uint64, vp, vp, C.c_size_t, vp]),
        "mc_rank_runs_forget": (C.c_int, [vp, C.c_uint64, vp]),
        "mc_rank_runs_check": (C.c_int, [vp, vp, u64p]),
        "mc_rank_runs_query": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, vp, vp, vp]),
        "mc_rank_runs_expand": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp]),
        "mc_rank_runs_debug_phases": (C.c_int, [C.c_int, u64p]),
        "mc_rank_runs_debug_groups": (C.c_int, [vp, C.c_uint64, u64p, vp]),
        "mc_rank_runs_guess_numbers": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, vp]),
        "mc_score_words_debug_kernel": (None, [C.c_int]),
        "mc_score_bpe_debug_kernel": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.restype, fn.argtypes = res, args
    # MC_HOST_SANITIZED=<libmi355mc_host_san.so> (python -m montecarlo_amd.build --sanitize; the process preloads the
    # sanitizer's runtime): the entry points that run on the host only -- table compiler, trainers, ingest, writers --
    # are taken from the library built under AddressSanitizer + UBSan (tests/test_sanitized_host.py)
    san = os.environ.get("MC_HOST_SANITIZED")
    if san:
        host = C.CDLL(san)
        taken = []
        failed_in = [lib]                # the library whose entry point failed last: each keeps its own message

        def tracked(fn, owner):
            def call(*a):
                rc = fn(*a)
                if isinstance(rc, int) and rc < 0:
                    failed_in[0] = owner
                return rc
            return call

        main_err = lib.mc_last_error
        for name, (res, args) in sig.items():
            if name in ("mc_last_error", "mc_abi_version"):
                continue
            if res is not C.c_int:
                if hasattr(host, name):
                    fn = getattr(host, name)
                    fn.restype, fn.argtypes = res, args
                    setattr(lib, name, fn)
                    taken.append(name)
                continue
            if hasattr(host, name):
                fn = getattr(host, name)
                fn.restype, fn.argtypes = res, args
                setattr(lib, name, tracked(fn, host))
                taken.append(name)
            else:
                setattr(lib, name, tracked(getattr(lib, name), lib))
        host.mc_last_error.restype, host.mc_last_error.argtypes = C.c_char_p, []
        lib.mc_last_error = lambda: host.mc_last_error() if failed_in[0] is host else main_err()
        lib.sanitized_host_entry_points = taken
    return lib


lib = _load()
EXPORTS = ["mc_abi_version", "mc_last_error", "mc_table_compile", "mc_table_free", "mc_table_array", "mc_table_info",
           "mc_ctx_hash", "mc_text_hash", "mc_score_words_workspace_bytes", "mc_score_words", "mc_score_bpe", "mc_train_nwords", "mc_train_backwords", "mc_train_v2", "mc_trained_free", "mc_counts_new", "mc_counts_free",
           "mc_counts_merge", "mc_counts_train", "mc_counts_import", "mc_counts_array", "mc_trained_array", "mc_trained_compile", "mc_lines_ingest", "mc_lines_free", "mc_lines_array", "mc_format_gc", "mc_format_gc_cps", "mc_model_upload", "mc_model_destroy", "mc_model_check", "mc_sample", "mc_sample_carry", "mc_sample_tune", "mc_sample_injected", "mc_sample_dedup",
           "mc_score", "mc_rank_workspace_bytes", "mc_rank_build", "mc_rank_check", "mc_rank_debug_fail", "mc_rank_debug_search", "mc_workspace_forget", "mc_rank_noalloc_max_keys", "mc_rank_build_noalloc", "mc_rank_query", "mc_order_workspace_bytes", "mc_order_keys", "mc_order_keys_begin", "mc_order_keys_end", "mc_guess_numbers_workspace_bytes", "mc_guess_numbers", "mc_comm_unique_id", "mc_comm_init", "mc_allreduce_rank", "mc_comm_destroy", "mc_enumerate", "mc_enum_info", "mc_enum_fetch",
           "mc_enum_free", "mc_rank_runs_workspace_bytes", "mc_rank_runs_capacity", "mc_rank_runs_max_keys", "mc_rank_runs_build",
           "mc_rank_runs_forget", "mc_rank_runs_check", "mc_rank_runs_query", "mc_rank_runs_expand", "mc_rank_runs_debug_phases",
           "mc_score_words_debug_kernel", "mc_score_bpe_debug_kernel",
           "mc_rank_runs_debug_groups", "mc_rank_runs_guess_numbers"]


def check(rc: int) -> int:
    if rc < 0:
        raise NativeError(rc, (lib.mc_last_error() or b"").decode("utf-8", "replace"))
    return rc
