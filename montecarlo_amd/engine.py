"""Device-side objects behind the lib4mc classes: a model resident on one GPU
and a rank index built from sampled keys.  torch is used for device memory and
streams only; every computation is a libmi355mc.so kernel on torch's current
stream.  No CPU fallback: constructing these without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np
import torch

from . import _native as nv
from .tables import FlatTable

DEFAULT_MAX_DRAWS = 1 << 16
CNT_NAMES = ("draws", "cdf_probes", "restarts", "failed", "score_probes", "score_steps", "indirect_draws", "score_too_long")


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("montecarlo_amd runs its hot path on an MI355X only: no GPU is visible and there is "
                           "no CPU fallback")


def _vp(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


class DeviceModel:
    def __init__(self, flat: FlatTable, device: Optional[int] = None):
        _require_gpu()
        self.flat = flat
        self.device = torch.cuda.current_device() if device is None else int(device)
        h = C.c_void_p()
        nv.check(nv.lib.mc_model_upload(flat.handle, self.device, C.byref(h)))
        self._h = h
        self.tdev = torch.device("cuda", self.device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and nv is not None and getattr(nv, "lib", None) is not None:   # (module may be gone at exit)
            nv.lib.mc_model_destroy(h)
            self._h = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.tdev).cuda_stream)

    def check(self) -> dict:
        """Synchronise, fetch-and-clear the device counters, raise if a sample diverged."""
        cnt = (C.c_uint64 * 8)()
        rc = nv.lib.mc_model_check(self._h, self._stream(), cnt)
        out = dict(zip(CNT_NAMES, list(cnt)))
        nv.check(rc)
        return out

    def tune(self, wg_per_cu: int = 0, skew: float = -1.0, hot_kb: int = -1):
        """Launch geometry of the sampling kernel (mc_sample_tune; never changes a sample); no arguments: defaults."""
        nv.check(nv.lib.mc_sample_tune(self._h, wg_per_cu, skew, hot_kb))

    def sample(self, count: int, seed: int, first_id: int = 0, max_draws: int = DEFAULT_MAX_DRAWS,
               want_len: bool = False, want_draws: bool = False, sym_stride: int = 0, out: torch.Tensor = None,
               carry=None):
        """Launch the sampling kernel; returns dict(ml2p=, len=, draws=, syms=) of device tensors.
        ``carry=(src, dst)``: the launch also copies the device tensor ``src`` into ``dst`` (a pinned host tensor
        as a rule) beside the sampling (mc_sample_carry); ``dst`` is complete when the launch is."""
        with torch.cuda.device(self.tdev):
            ml2p = out if out is not None else torch.empty(count, dtype=torch.float64, device=self.tdev)
            assert ml2p.numel() >= count and ml2p.dtype == torch.float64 and ml2p.is_contiguous()
            ln = torch.empty(count, dtype=torch.int32, device=self.tdev) if (want_len or sym_stride) else None
            dr = torch.empty(count, dtype=torch.int32, device=self.tdev) if want_draws else None
            sy = torch.zeros((count, sym_stride), dtype=torch.int32, device=self.tdev) if sym_stride else None
            if carry is not None:
                src, dst = carry
                nbytes = src.numel() * src.element_size()
                assert src.is_contiguous() and dst.is_contiguous() and dst.numel() * dst.element_size() == nbytes
                nv.check(nv.lib.mc_sample_carry(self._h, first_id, count, seed & 0xFFFFFFFFFFFFFFFF, max_draws,
                                                _vp(ml2p), _vp(ln), _vp(dr), _vp(sy), sym_stride, _vp(src), _vp(dst),
                                                nbytes, self._stream()))
                return {"ml2p": ml2p, "len": ln, "draws": dr, "syms": sy}
            nv.check(nv.lib.mc_sample(self._h, first_id, count, seed & 0xFFFFFFFFFFFFFFFF, max_draws, _vp(ml2p),
                                      _vp(ln), _vp(dr), _vp(sy), sym_stride, self._stream()))
        return {"ml2p": ml2p, "len": ln, "draws": dr, "syms": sy}

    def dedup(self, syms: torch.Tensor, lens: torch.Tensor, count: int):
        """GPU-side `sampled_pwds`: -> (first sample index of every distinct row, ascending; its
        occurrences; indices of rows longer than the stride, which the caller replays)."""
        stride = syms.shape[1]
        cap = 1 << max(4, (2 * count - 1).bit_length())
        with torch.cuda.device(self.tdev):
            key = torch.empty(cap, dtype=torch.int64, device=self.tdev)
            first = torch.empty(cap, dtype=torch.int64, device=self.tdev)
            cnt = torch.empty(cap, dtype=torch.int32, device=self.tdev)
            bad = torch.empty(1, dtype=torch.int32, device=self.tdev)
            nv.check(nv.lib.mc_sample_dedup(_vp(syms), _vp(lens), stride, count, cap, _vp(key), _vp(first), _vp(cnt),
                                            _vp(bad), self._stream()))
            if int(bad.item()):
                raise RuntimeError(f"{int(bad.item())} sampled strings collide in the 64-bit row hash")
            slots = torch.nonzero(cnt).flatten()
            f, c = first[slots], cnt[slots].to(torch.int64)
            order = torch.argsort(f)
            long_ids = torch.nonzero(lens[:count] > stride).flatten()
        return f[order], c[order], long_ids

    def sample_injected(self, uniforms, want_draws: bool = True, sym_stride: int = 0):
        """Parity mode: ``uniforms[i, k]`` is the k-th uniform of sample i."""
        u = torch.as_tensor(np.ascontiguousarray(uniforms, dtype=np.float64)) if not torch.is_tensor(uniforms) \
            else uniforms
        u = u.to(self.tdev, dtype=torch.float64).contiguous()
        count, dps = u.shape
        with torch.cuda.device(self.tdev):
            ml2p = torch.empty(count, dtype=torch.float64, device=self.tdev)
            ln = torch.empty(count, dtype=torch.int32, device=self.tdev)
            dr = torch.empty(count, dtype=torch.int32, device=self.tdev) if want_draws else None
            sy = torch.zeros((count, sym_stride), dtype=torch.int32, device=self.tdev) if sym_stride else None
            nv.check(nv.lib.mc_sample_injected(self._h, count, _vp(u), dps, _vp(ml2p), _vp(ln), _vp(dr), _vp(sy),
                                               sym_stride, self._stream()))
            torch.cuda.current_stream(self.tdev).synchronize()   # `u` must outlive the kernel
        return {"ml2p": ml2p, "len": ln, "draws": dr, "syms": sy}

    def score_ids(self, syms: np.ndarray, off: np.ndarray, out: torch.Tensor = None) -> torch.Tensor:
        count = len(off) - 1
        with torch.cuda.device(self.tdev):
            d_syms = torch.from_numpy(np.ascontiguousarray(syms, dtype=np.uint32).view(np.int32)).to(self.tdev)
            d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.int64)).to(self.tdev)
            return self.score_device(d_syms, d_off, count, out)

    def score_device(self, d_syms: torch.Tensor, d_off: torch.Tensor, count: int, out: torch.Tensor = None):
        with torch.cuda.device(self.tdev):
            res = out if out is not None else torch.empty(count, dtype=torch.float64, device=self.tdev)
            nv.check(nv.lib.mc_score(self._h, _vp(d_syms), _vp(d_off), count, _vp(res), self._stream()))
        return res


    def score_words(self, cps: np.ndarray, off: np.ndarray, max_len: int, iter_max: int, want_split: bool = True):
        """calc_ml2p as the DFS over segmentations (any alphabet): (ml2p tensor, split masks or None)."""
        count = len(off) - 1
        with torch.cuda.device(self.tdev):
            d_cps = torch.from_numpy(np.ascontiguousarray(cps, dtype=np.uint32).view(np.int32)).to(self.tdev)
            d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.int64)).to(self.tdev)
            res = torch.empty(count, dtype=torch.float64, device=self.tdev)
            words = (max_len + 1 + 63) // 64 if want_split else 0
            split = torch.empty((count, words), dtype=torch.int64, device=self.tdev) if want_split else None
            need = nv.lib.mc_score_words_workspace_bytes(self._h, count, max_len)
            ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.tdev)
            nv.check(nv.lib.mc_score_words(self._h, _vp(d_cps), _vp(d_off), count, max_len,
                                           min(int(iter_max), 0xFFFFFFFFFFFFFFFF) if iter_max > 0 else 0, _vp(res),
                                           _vp(split), words, _vp(ws), ws.numel(), self._stream()))
            torch.cuda.current_stream(self.tdev).synchronize()      # inputs and workspace must outlive the kernel
        return res, split


    def score_words_device(self, d_cps: torch.Tensor, d_off: torch.Tensor, count: int, max_len: int, iter_max: int,
                           out: torch.Tensor, workspace: torch.Tensor, split: torch.Tensor = None):
        """mc_score_words on inputs, output and workspace (mc_score_words_workspace_bytes) that already live on the
        GPU: nothing is copied, allocated or waited for."""
        with torch.cuda.device(self.tdev):
            nv.check(nv.lib.mc_score_words(self._h, _vp(d_cps), _vp(d_off), count, max_len,
                                           min(int(iter_max), 0xFFFFFFFFFFFFFFFF) if iter_max > 0 else 0, _vp(out),
                                           _vp(split), split.shape[1] if split is not None else 0, _vp(workspace),
                                           workspace.numel(), self._stream()))
        return out

    def score_bpe_device(self, d_cps: torch.Tensor, d_cls: torch.Tensor, d_off: torch.Tensor, count: int,
                         out: torch.Tensor):
        """mc_score_bpe on resident inputs and output."""
        with torch.cuda.device(self.tdev):
            nv.check(nv.lib.mc_score_bpe(self._h, _vp(d_cps), _vp(d_cls), _vp(d_off), count, _vp(out), self._stream()))
        return out

    def score_bpe(self, cps: np.ndarray, cls: np.ndarray, off: np.ndarray) -> torch.Tensor:
        count = len(off) - 1
        with torch.cuda.device(self.tdev):
            d_cps = torch.from_numpy(np.ascontiguousarray(cps, dtype=np.uint32).view(np.int32)).to(self.tdev)
            d_cls = torch.from_numpy(np.ascontiguousarray(cls, dtype=np.uint8)).to(self.tdev)
            d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.int64)).to(self.tdev)
            res = torch.empty(count, dtype=torch.float64, device=self.tdev)
            nv.check(nv.lib.mc_score_bpe(self._h, _vp(d_cps), _vp(d_cls), _vp(d_off), count, _vp(res), self._stream()))
            torch.cuda.current_stream(self.tdev).synchronize()      # inputs must outlive the kernel
        return res


    def enumerate(self, threshold: float, min_len: int, max_len: int, max_nodes: int = 1 << 31):
        """Threshold enumeration (mc_enumerate): (ml2p, lengths, symbol rows) of every string with
        -log2 p < threshold, as numpy arrays in the reference's depth-first order."""
        h = C.c_void_p()
        with torch.cuda.device(self.tdev):
            nv.check(nv.lib.mc_enumerate(self._h, float(threshold), min_len, max_len, max_nodes, self._stream(),
                                         C.byref(h)))
        try:
            info = (C.c_uint64 * 4)()
            nv.check(nv.lib.mc_enum_info(h, info))
            n, stride = int(info[0]), int(info[1])
            ml2p = np.empty(n, dtype=np.float64)
            lens = np.empty(n, dtype=np.uint32)
            syms = np.empty((n, stride), dtype=np.uint32)
            for what, arr in ((nv.ENUM_ML2P, ml2p), (nv.ENUM_LEN, lens), (nv.ENUM_SYMS, syms)):
                if arr.nbytes:
                    nv.check(nv.lib.mc_enum_fetch(h, what, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
            self.last_enum_info = {"strings": n, "stride": stride, "nodes": int(info[2]), "levels": int(info[3])}
        finally:
            nv.lib.mc_enum_free(h)
        return ml2p, lens, syms


class RunsBuffers:
    """Outputs and workspace of one run-length rank build of up to ``n`` keys (mc_rank_runs_build), for callers that
    build again and again without allocating: ``DeviceRank(keys, runs=buffers)``."""

    def __init__(self, n: int, device):
        self.n = int(n)
        self.tdev = torch.device(device)
        with torch.cuda.device(self.tdev):
            self.capacity = int(nv.lib.mc_rank_runs_capacity(self.n))
            self.vals = torch.empty(self.capacity, dtype=torch.float64, device=self.tdev)
            self.cum_count = torch.empty(self.capacity, dtype=torch.int64, device=self.tdev)
            self.cum_pos = torch.empty(self.capacity, dtype=torch.float64, device=self.tdev)
            self.info = torch.zeros(8, dtype=torch.int64, device=self.tdev)
            self.ws = torch.empty(max(1, nv.lib.mc_rank_runs_workspace_bytes(self.n)), dtype=torch.uint8, device=self.tdev)
            # (memory fresh from the allocator may hold anything, another build's hint included: start clean)
            nv.check(nv.lib.mc_rank_runs_forget(_vp(self.ws), self.n, C.c_void_p(torch.cuda.current_stream(self.tdev).cuda_stream)))


# The size from which this process has seen the run index refuse its keys (a refused build costs its 0.1-1 ms and
# the expanded build after it): later default builds of that many keys or more take mc_rank_build straight away.
# Models differ -- 1e7 samples of the headline model hold 170,000 distinct values and suit the index up to 1e7
# keys, those of the v2 6-gram model up to 4e6, those of a sub-word model a million values and do not
# (scripts/runs_sweep.py, profiles/r03_d_runs_sweep.txt); nothing but a build tells.  `forget_rank_choices()` starts over.
_PREFER_SORTED_FROM = [None]


def forget_rank_choices():
    _PREFER_SORTED_FROM[0] = None


def runs_index_wanted(n: int) -> bool:
    """The run-length index is the default wherever it applies; MC_RANK_INDEX=sorted keeps the expanded arrays."""
    import os
    if _PREFER_SORTED_FROM[0] is not None and n >= _PREFER_SORTED_FROM[0]:
        return False
    return 0 < n <= int(nv.lib.mc_rank_runs_max_keys()) and os.environ.get("MC_RANK_INDEX", "runs") != "sorted"


class DeviceRank:
    """The rank index of sampled keys on one GPU (MonteCarloLib.__init__, lib4mc/MonteCarloLib.py:10-28).

    ``index="runs"`` (the default where it applies): one entry per distinct value -- value, samples at or below
    it, summed weights at or below it (mc_rank_runs_build) -- which is all ``bisect_right`` ever reads; the sorted
    keys and the positions are written only when somebody looks at ``keys_sorted`` / ``positions``.  Keys that do
    not repeat do not suit that index (mc_rank_runs_check says so): the build then falls back to
    ``index="sorted"``, the expanded arrays of mc_rank_build, which sort anything, and the process remembers the size
    (``runs_index_wanted``): later default builds of that many keys or more go there directly.  A caller that brings its own
    buffers (``runs=`` or ``keys_sorted=`` / ``positions=``) decides the kind by that and checks the build itself
    (``check()``): nothing waits for the GPU here."""

    def __init__(self, keys: torch.Tensor, n_global: Optional[int] = None, workspace: torch.Tensor = None,
                 keys_sorted: torch.Tensor = None, positions: torch.Tensor = None, runs: "RunsBuffers" = None,
                 index: str = None):
        _require_gpu()
        assert keys.is_cuda and keys.dtype == torch.float64 and keys.is_contiguous()
        self.n = keys.numel()
        self.n_global = self.n if n_global is None else int(n_global)
        self.tdev = keys.device
        self._keys_in = keys
        self._sorted = keys_sorted
        self._positions = positions
        self._ws = workspace
        self.runs = None
        own_buffers = keys_sorted is not None or positions is not None or workspace is not None
        if index is None:
            index = "runs" if runs is not None or (not own_buffers and runs_index_wanted(self.n)) else "sorted"
        if index == "runs" and self.n:
            self.runs = runs if runs is not None else RunsBuffers(self.n, self.tdev)
            assert self.runs.n >= self.n and self.runs.tdev == self.tdev
            self._build_runs()
            if runs is None:                                  # the library's own buffers: settle the kind now
                rc = self._runs_status()[0]
                if rc == nv.ERR_UNSUITED:
                    prev = _PREFER_SORTED_FROM[0]             # (this size and larger: the expanded arrays from now on)
                    _PREFER_SORTED_FROM[0] = self.n if prev is None else min(prev, self.n)
                    self.runs = None                          # keys that do not repeat: the expanded arrays sort anything
                else:
                    nv.check(rc)
        if self.runs is None:
            self._build_sorted()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.tdev).cuda_stream)

    def _build_runs(self):
        r = self.runs
        with torch.cuda.device(self.tdev):
            nv.check(nv.lib.mc_rank_runs_build(_vp(self._keys_in), self.n, self.n_global, _vp(r.vals), _vp(r.cum_count),
                                               _vp(r.cum_pos), r.capacity, _vp(r.info), _vp(r.ws), r.ws.numel(),
                                               self._stream()))

    def _runs_status(self):
        h = (C.c_uint64 * 8)()
        with torch.cuda.device(self.tdev):
            rc = nv.lib.mc_rank_runs_check(_vp(self.runs.info), self._stream(), h)
        return rc, list(h)

    def _build_sorted(self):
        keys = self._keys_in
        with torch.cuda.device(self.tdev):
            need = nv.lib.mc_rank_workspace_bytes(self.n)
            fresh = self._ws is None
            ws = self._ws if self._ws is not None else torch.empty(need, dtype=torch.uint8, device=self.tdev)
            if self._sorted is None:
                self._sorted = torch.empty_like(keys)
            if self._positions is None:
                self._positions = torch.empty_like(keys)
            if fresh and self.n:
                # (a caller's workspace carries the list of heavy values from one build to the next, include/mi355mc.h;
                # memory fresh from the allocator may hold anything, another build's list included: start clean)
                nv.check(nv.lib.mc_workspace_forget(_vp(ws), self.n, self._stream()))
            nv.check(nv.lib.mc_rank_build(_vp(keys), self.n, self.n_global, _vp(self._sorted), _vp(self._positions),
                                          _vp(ws), ws.numel(), self._stream()))
            self._ws = ws

    def _expand(self):
        """The reference's two arrays from the runs, for callers that look at them."""
        if self._sorted is None or self._positions is None:
            with torch.cuda.device(self.tdev):
                srt = torch.empty(self.n, dtype=torch.float64, device=self.tdev)
                pos = torch.empty(self.n, dtype=torch.float64, device=self.tdev)
                r = self.runs
                nv.check(nv.lib.mc_rank_runs_expand(_vp(r.vals), _vp(r.cum_count), _vp(r.cum_pos), _vp(r.info), self.n,
                                                    self.n_global, _vp(srt), _vp(pos), self._stream()))
            self._sorted, self._positions = srt, pos

    @property
    def keys_sorted(self) -> torch.Tensor:
        if self.runs is not None:
            self._expand()
        return self._sorted

    @property
    def positions(self) -> torch.Tensor:
        if self.runs is not None:
            self._expand()
        return self._positions

    def check(self) -> dict:
        """Synchronise and read the build's status: raises if it failed on the device (positions are NaN then);
        returns the build's counters."""
        if self.runs is not None:
            rc, h = self._runs_status()
            out = dict(zip(("distinct_values", "error", "keys_counted", "heavy_values", "tail_keys", "list_age"), h))
            out["index"] = "runs"
            nv.check(rc)
            return out
        info = (C.c_uint64 * 8)()
        with torch.cuda.device(self.tdev):
            rc = nv.lib.mc_rank_check(_vp(self._ws), self.n, self._stream(), info)
        out = dict(zip(("error", "heavy_values", "tail_keys", "levels", "finish_tiles", "segments", "list_age"), list(info)))
        out["index"] = "sorted"
        nv.check(rc)
        return out

    def query(self, x: torch.Tensor, out_pos: torch.Tensor = None, out_cnt: torch.Tensor = None):
        """(pos, cnt): cnt = bisect_right(sorted, x), pos = positions[cnt-1] (0.0 when cnt == 0)."""
        assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
        t = x.numel()
        with torch.cuda.device(self.tdev):
            pos = out_pos if out_pos is not None else torch.empty(t, dtype=torch.float64, device=self.tdev)
            cnt = out_cnt if out_cnt is not None else torch.empty(t, dtype=torch.int64, device=self.tdev)
            if self.runs is not None:
                r = self.runs
                nv.check(nv.lib.mc_rank_runs_query(_vp(r.vals), _vp(r.cum_count), _vp(r.cum_pos), _vp(r.info), _vp(x), t,
                                                   _vp(pos), _vp(cnt), self._stream()))
            elif self.n == 0:
                pos.zero_()
                cnt.zero_()
            else:
                nv.check(nv.lib.mc_rank_query(_vp(self._sorted), _vp(self._positions), self.n, _vp(x), t, _vp(pos),
                                              _vp(cnt), self._stream()))
        return pos, cnt


def device_order(x: torch.Tensor, out_sorted: torch.Tensor = None, out_order: torch.Tensor = None,
                 workspace: torch.Tensor = None):
    """Stable ascending order of the keys ``x`` (device float64): (sorted keys, order as int32 holding uint32
    indices) -- mc_order_keys, what ``sorted(..., key=ml2p)`` of parse_file computes on the host.  With a
    ``workspace`` of the caller's (mc_order_workspace_bytes) nothing is allocated and nothing waited for: the
    caller checks ``order[0] != -1`` (the device's failure mark) once it has the result."""
    _require_gpu()
    assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
    t = x.numel()
    with torch.cuda.device(x.device):
        srt = out_sorted if out_sorted is not None else torch.empty_like(x)
        order = out_order if out_order is not None else torch.empty(t, dtype=torch.int32, device=x.device)
        if t:
            ws = workspace if workspace is not None else torch.empty(nv.lib.mc_order_workspace_bytes(t),
                                                                     dtype=torch.uint8, device=x.device)
            if workspace is None:                            # (no list of another sort's heavy values: see DeviceRank)
                nv.check(nv.lib.mc_workspace_forget(_vp(ws), t, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
            nv.check(nv.lib.mc_order_keys(_vp(x), t, _vp(srt), _vp(order), _vp(ws), ws.numel(),
                                          C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
            if workspace is None:
                torch.cuda.current_stream(x.device).synchronize()    # the workspace must outlive the kernels
                if int(order[0]) == -1:
                    raise nv.NativeError("mc_order_keys: the key sort reported an error on the device (no order "
                                         "produced)")
    return srt, order


def device_order_begin(x: torch.Tensor, out_sorted: torch.Tensor, out_values: torch.Tensor, workspace: torch.Tensor):
    """The first half of device_order (mc_order_keys_begin): the key sort; ``out_values`` receives the scores in
    ascending order as numbers -- all a bisection needs -- while ``out_sorted`` and the order wait for
    device_order_end (same tensors, same workspace; typically on another stream than the queries)."""
    _require_gpu()
    assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
    assert out_values.dtype == torch.float64 and out_values.numel() == x.numel() and out_values.is_contiguous()
    with torch.cuda.device(x.device):
        if x.numel():
            nv.check(nv.lib.mc_order_keys_begin(_vp(x), x.numel(), _vp(out_sorted), _vp(out_values), _vp(workspace),
                                                workspace.numel(),
                                                C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return out_values


def device_order_end(x: torch.Tensor, out_sorted: torch.Tensor, out_order: torch.Tensor, workspace: torch.Tensor):
    """The second half (mc_order_keys_end): the passwords' stable order and the sorted scores as the caller's bits."""
    _require_gpu()
    with torch.cuda.device(x.device):
        if x.numel():
            nv.check(nv.lib.mc_order_keys_end(_vp(x), x.numel(), _vp(out_sorted), _vp(out_order), _vp(workspace),
                                              workspace.numel(),
                                              C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return out_sorted, out_order


def device_query_guess_numbers(rk: "DeviceRank", x: torch.Tensor, add1: bool = True, out_pos: torch.Tensor = None,
                               out_cnt: torch.Tensor = None, out: torch.Tensor = None, workspace: torch.Tensor = None,
                               exact: torch.Tensor = None):
    """rk.query(x) and device_guess_numbers in one call (mc_rank_runs_guess_numbers: two launches, not three) when
    ``rk`` holds a run index; otherwise the two calls.  -> (pos, cnt, ranks, exact)."""
    _require_gpu()
    assert x.is_cuda and x.dtype == torch.float64 and x.is_contiguous()
    t = x.numel()
    dev = x.device
    with torch.cuda.device(dev):
        pos = out_pos if out_pos is not None else torch.empty(t, dtype=torch.float64, device=dev)
        cnt = out_cnt if out_cnt is not None else torch.empty(t, dtype=torch.int64, device=dev)
        if rk.runs is None:
            rk.query(x, out_pos=pos, out_cnt=cnt)
            ranks, ex, _ = device_guess_numbers(pos, cnt, add1, out=out, workspace=workspace, exact=exact)
            return pos, cnt, ranks, ex
        ranks = out if out is not None else torch.empty(t, dtype=torch.int64, device=dev)
        ex = exact if exact is not None else torch.empty(1, dtype=torch.int64, device=dev)
        ws = workspace if workspace is not None else torch.empty(nv.lib.mc_guess_numbers_workspace_bytes(t),
                                                                 dtype=torch.uint8, device=dev)
        r = rk.runs
        nv.check(nv.lib.mc_rank_runs_guess_numbers(_vp(r.vals), _vp(r.cum_count), _vp(r.cum_pos), _vp(r.info), _vp(x), t,
                                                   1 if add1 else 0, _vp(pos), _vp(cnt), _vp(ranks), _vp(ex), _vp(ws),
                                                   ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return pos, cnt, ranks, ex


def device_guess_numbers(pos: torch.Tensor, cnt: Optional[torch.Tensor], add1: bool = True,
                         out: torch.Tensor = None, workspace: torch.Tensor = None, exact: torch.Tensor = None):
    """mc_guess_numbers on device vectors in ascending order of -log2 p: (ranks int64, exact) where ``exact`` is a
    one-element device tensor = the number of leading ranks that are exact (the rest exceed int64)."""
    t = pos.numel()
    with torch.cuda.device(pos.device):
        ranks = out if out is not None else torch.empty(t, dtype=torch.int64, device=pos.device)
        ex = exact if exact is not None else torch.empty(1, dtype=torch.int64, device=pos.device)
        ws = workspace if workspace is not None else torch.empty(nv.lib.mc_guess_numbers_workspace_bytes(t),
                                                                 dtype=torch.uint8, device=pos.device)
        nv.check(nv.lib.mc_guess_numbers(_vp(pos), _vp(cnt), t, 1 if add1 else 0, _vp(ranks), _vp(ex), _vp(ws),
                                         ws.numel(), C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)))
    return ranks, ex, ws


def log2_n(n: int) -> float:
    return math.log2(n)
