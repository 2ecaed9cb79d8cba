"""Multi-GPU decomposition of the guess-number estimate.

positions[idx-1] = sum over samples j with ml2p_j <= x of 2^(ml2p_j - log2 N)
(reference lib4mc/MonteCarloLib.py:24-32) is a sum over samples, so it splits over
disjoint sample shards as long as every shard weights with the GLOBAL N.  Each
rank draws its own sample ids on the shared Philox stream, sorts and scans
locally and answers every test password locally; ONE all-reduce (SUM) of the
fp64 partial sums finishes the estimate (`combine_positions`; the int64 counts that
decide the reference's ``idx > 0`` branch can be summed too, `combine_partials`, but
are implied by the positions).  Samples never move between GPUs.

Backend: torch.distributed -- "nccl" is RCCL over xGMI on the GPU box, "gloo" in
the CPU tests.  T <= 1e6 test passwords is 8 MB: latency-bound, one bucket.

`use_group` makes the sharding ambient for the lib4mc classes: ``MonteCarlo.sample`` then draws only
this rank's share of the sample ids (the ids are global, so the sample SET is the one a single GPU
draws), ``sampled_pwds`` is merged over the ranks in global first-sampled order, and
``MonteCarloLib`` all-reduces its queries -- a driver written against the reference's surface
(backwords_secondary_main.py, BASELINE config 5) runs unchanged on every rank.
"""
from __future__ import annotations

import random
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist

_ACTIVE = None      # the process group MonteCarlo.sample / MonteCarloLib shard over (None: one GPU)


def use_group(group=True):
    """Shard every later ``sample`` / ``MonteCarloLib`` over ``group`` (True: the default group; None:
    back to one GPU).  The ranks must make the same calls in the same order; their ``random`` states
    are made equal here (rank 0's), because the reference's drivers draw the Philox seed and
    ``random.sample`` from it."""
    global _ACTIVE
    if group is None or group is False:
        _ACTIVE = None
        return
    _ACTIVE = dist.group.WORLD if group is True else group
    if dist.get_world_size(_ACTIVE) > 1:
        box = [random.getstate() if dist.get_rank(_ACTIVE) == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(_ACTIVE, 0), group=_ACTIVE)
        random.setstate(box[0])


def active_group():
    """The ambient group if it has more than one rank, else None."""
    if _ACTIVE is not None and dist.is_initialized() and dist.get_world_size(_ACTIVE) > 1:
        return _ACTIVE
    return None


def merge_distinct(local: Sequence[Tuple[str, float, int]], group) -> List[Tuple[str, float, int]]:
    """Every rank's distinct sampled strings ``(string, ml2p, occurrences)`` -- each list in its shard's
    first-sampled order, shards being contiguous id ranges in rank order -- as ONE list in global
    first-sampled order with the occurrences summed: what ``sample(size, sampled_pwds)`` builds on a
    single GPU (lib4mc/MonteCarloParent.py:39-45), on every rank."""
    parts: List = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, list(local), group=group)
    seen: Dict[str, List] = {}
    for part in parts:
        for pwd, x, c in part:
            hit = seen.get(pwd)
            if hit is None:
                seen[pwd] = [x, c]
            else:
                hit[1] += c
    return [(pwd, x, c) for pwd, (x, c) in seen.items()]


def shard_range(n_global: int, rank: int, world: int) -> Tuple[int, int]:
    """Sample ids [first, first+count) of ``rank``: contiguous, sizes differ by at most one."""
    base, extra = divmod(n_global, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def combine_partials(pos: torch.Tensor, cnt: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sum the per-shard (partial position, count) vectors in place over the group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(pos, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    return pos, cnt


def combine_positions(pos: torch.Tensor, group=None) -> torch.Tensor:
    """The one collective of the hot path: sum the per-shard partial positions in place.  The counts
    need not travel: every weight 2^(ml2p - log2 N) is positive, so "some shard holds a sample
    <= x" (the reference's ``idx > 0``) is the same as "the summed position is > 0"."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(pos, op=dist.ReduceOp.SUM, group=group)
    return pos


class NativeComm:
    """The same sum through the C ABI (mc_comm_init / mc_allreduce_rank: RCCL without torch.distributed), for
    callers that bind the library from another host language.  ``unique_id()`` on rank 0, its 128 bytes carried
    to the other ranks by whatever the caller has, then every rank constructs with its GPU current."""

    COMM_ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from . import _native as nv
        buf = (C.c_ubyte * NativeComm.COMM_ID_BYTES)()
        nv.check(nv.lib.mc_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id: bytes, rank: int, world: int):
        import ctypes as C

        from . import _native as nv
        if len(unique_id) != self.COMM_ID_BYTES:
            raise ValueError(f"the communicator id has {self.COMM_ID_BYTES} bytes, got {len(unique_id)}")
        self._nv, self._h = nv, C.c_void_p()
        nv.check(nv.lib.mc_comm_init(unique_id, rank, world, C.byref(self._h)))
        self.rank, self.world = rank, world

    def combine(self, pos: torch.Tensor, cnt: torch.Tensor = None) -> torch.Tensor:
        import ctypes as C
        assert pos.is_cuda and pos.dtype == torch.float64 and pos.is_contiguous()
        assert cnt is None or (cnt.is_cuda and cnt.dtype == torch.int64 and cnt.is_contiguous() and cnt.numel() == pos.numel())
        self._nv.check(self._nv.lib.mc_allreduce_rank(
            self._h, C.c_void_p(pos.data_ptr()), C.c_void_p(cnt.data_ptr()) if cnt is not None else None,
            pos.numel(), C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)))
        return pos

    def close(self):
        if self._h:
            self._nv.lib.mc_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def raise_on_poisoned(pos: torch.Tensor) -> None:
    """A rank build that failed on ANY shard poisons that shard's positions with NaN (csrc/mc_rank.hip: error
    word); NaN survives the SUM, so every rank sees it here.  It must not be read as "no sample counted"
    (NaN > 0 is False, which would turn every guess number into 1)."""
    if bool(torch.isnan(pos).any()):
        raise RuntimeError("rank estimation failed on at least one shard: the summed positions contain NaN "
                           "(mc_rank_build reported an error on its device, or a query key was NaN)")


def estimates_from_positions(pos: torch.Tensor) -> torch.Tensor:
    """``positions[idx-1] if idx > 0 else 1`` from the summed positions alone."""
    raise_on_poisoned(pos)
    return torch.where(pos > 0, pos, torch.ones_like(pos))


def estimates(pos: torch.Tensor, cnt: torch.Tensor) -> torch.Tensor:
    """The reference's ``positions[idx-1] if idx > 0 else 1`` on combined vectors."""
    return torch.where(cnt > 0, pos, torch.ones_like(pos))


def sharded_guess_numbers(model, n_global: int, seed: int, test_ml2p: torch.Tensor, group=None,
                          first_id: int = 0):
    """One rank's part of: sample n_global strings over the group, build the local rank
    index, query the (replicated) test keys, all-reduce.  Returns (estimates, cnt, DeviceRank)."""
    from .engine import DeviceRank
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count = shard_range(n_global, rank, world)
    keys = model._dev.sample(count, seed, first_id=first_id + first, max_draws=model.max_draws)["ml2p"]
    rk = DeviceRank(keys, n_global=n_global)
    pos, cnt = rk.query(test_ml2p)
    combine_partials(pos, cnt, group)
    model._dev.check()
    return estimates(pos, cnt), cnt, rk
