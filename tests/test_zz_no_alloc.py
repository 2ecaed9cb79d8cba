"""mc_rank_build_noalloc (include/mi355mc.h): the rank build with its workspace in the library's own
device array, for callers that cannot allocate.  Called through the C ABI and compared bit for bit with
mc_rank_build (caller workspace) on two calls in a row, the second with a smaller input; collected
last (test_zz_*) so that a failure here hides no other test under `pytest -x`."""
import ctypes as C

import numpy as np
import pytest
import torch

from montecarlo_amd import _native as nv

pytestmark = pytest.mark.gpu


def _vp(t):
    return C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _build_with_workspace(keys, n_global):
    dev = keys.device
    n = keys.numel()
    ws = torch.empty(max(1, nv.lib.mc_rank_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    srt, pos = torch.empty_like(keys), torch.empty_like(keys)
    nv.check(nv.lib.mc_rank_build(_vp(keys), n, n_global, _vp(srt), _vp(pos), _vp(ws), ws.numel(), _stream(dev)))
    torch.cuda.synchronize()
    return srt, pos


def _build_noalloc(keys, n_global, srt, pos):
    """Outputs are preallocated by the caller: nothing is allocated between here and the result."""
    dev = keys.device
    nv.check(nv.lib.mc_rank_build_noalloc(_vp(keys), keys.numel(), n_global, _vp(srt), _vp(pos), _stream(dev)))
    torch.cuda.synchronize()


def test_zz_rank_build_noalloc_equals_rank_build():
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(20250304)
    big_n, small_n = 1_000_003, 70_001
    assert big_n <= nv.lib.mc_rank_noalloc_max_keys()
    # keys shaped like sampled -log2 p: a few heavy repeated values and a continuous tail
    heavy = torch.tensor([17.25, 21.5, 33.0078125, 40.0], dtype=torch.float64)
    pick = torch.randint(0, 8, (big_n,), generator=g)
    tail = torch.rand(big_n, generator=g, dtype=torch.float64) * 70 + 3
    big = torch.where(pick < 4, heavy[pick.clamp(max=3)], tail).to(dev).contiguous()
    small = (torch.rand(small_n, generator=g, dtype=torch.float64) * 50 + 5).to(dev).contiguous()
    out = {n: (torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.float64, device=dev))
           for n in (big_n, small_n)}
    want_big = _build_with_workspace(big, big_n)
    want_small = _build_with_workspace(small, 4 * small_n)
    # two calls in a row on the same GPU's array, the second smaller: nothing of the first may leak
    _build_noalloc(big, big_n, *out[big_n])
    _build_noalloc(small, 4 * small_n, *out[small_n])
    for (srt, pos), (wsrt, wpos), keys in ((out[big_n], want_big, big), (out[small_n], want_small, small)):
        assert torch.equal(srt.view(torch.int64), wsrt.view(torch.int64))
        assert torch.equal(pos.view(torch.int64), wpos.view(torch.int64))
        assert torch.equal(srt.view(torch.int64), torch.sort(keys)[0].view(torch.int64))
        assert bool(torch.isfinite(pos).all())
    # and once more the large one (the array now holds the small call's leftovers)
    _build_noalloc(big, big_n, *out[big_n])
    assert torch.equal(out[big_n][0].view(torch.int64), want_big[0].view(torch.int64))
    assert torch.equal(out[big_n][1].view(torch.int64), want_big[1].view(torch.int64))


def test_zz_rank_build_noalloc_rejects_sizes_beyond_its_scratch():
    dev = torch.device("cuda", torch.cuda.current_device())
    limit = int(nv.lib.mc_rank_noalloc_max_keys())
    n = limit + 1
    keys = torch.zeros(n, dtype=torch.float64, device=dev)
    srt, pos = torch.empty_like(keys), torch.empty_like(keys)
    rc = nv.lib.mc_rank_build_noalloc(_vp(keys), n, n, _vp(srt), _vp(pos), _stream(dev))
    assert rc == -1
    msg = nv.lib.mc_last_error().decode()
    assert "at most" in msg and str(limit) in msg
    # the largest accepted size works
    keys = (torch.arange(limit, dtype=torch.float64, device=dev) % 977) * 0.0625 + 1.0
    srt, pos = torch.empty_like(keys), torch.empty_like(keys)
    _build_noalloc(keys, limit, srt, pos)
    assert torch.equal(srt, torch.sort(keys)[0])
    assert np.isfinite(float(pos[-1]))
