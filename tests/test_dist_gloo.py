"""The N>1 path on CPU: two gloo ranks each hold one shard of the samples (drawn and
ranked by the C oracle, since there is no GPU here), the PRODUCT's sharding and
combination code (montecarlo_amd/dist.py) merges the per-shard partial sums, and the
result must equal the single-shard estimate of the whole sample set (1e-9 relative)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden_lines


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_global, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from montecarlo_amd import dist as mcd
    from oracle import c_oracle, mc_oracle as orc
    probs, _ = orc.train_backwords(golden_lines("train_char.txt"), "", "\x00", "\x03", 0, 1, 10, 256)
    cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_BACKOFF, start_units=["\x00"])
    first, count = mcd.shard_range(n_global, rank, world)
    keys = cm.sample(count, 77, first_id=first)["ml2p"]
    srt, pos = c_oracle.rank_build(keys, n_global=n_global)
    x = np.linspace(0.0, 60.0, 501)
    p, c = c_oracle.rank_query(srt, pos, x)
    tp, tc = torch.from_numpy(p.copy()), torch.from_numpy(c.astype(np.int64))
    only_pos = mcd.estimates_from_positions(mcd.combine_positions(tp.clone()))
    mcd.combine_partials(tp, tc)
    est = mcd.estimates(tp, tc)
    assert torch.equal(est, only_pos)                          # the counts are implied by the positions
    if rank == 0:
        np.save(os.path.join(out_dir, "est.npy"), est.numpy())
        np.save(os.path.join(out_dir, "cnt.npy"), tc.numpy())
    dist.destroy_process_group()


def test_shard_range_partitions():
    from montecarlo_amd.dist import shard_range
    for n, w in ((10, 3), (7, 8), (10_000_000_000, 8), (0, 2)):
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n
        for (f0, c0), (f1, _) in zip(parts, parts[1:]):
            assert f0 + c0 == f1
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_shard(tmp_path):
    n_global, world = 5001, 2
    mp.spawn(_worker, args=(world, _free_port(), n_global, str(tmp_path)), nprocs=world, join=True)
    est = np.load(tmp_path / "est.npy")
    cnt = np.load(tmp_path / "cnt.npy")
    from oracle import c_oracle, mc_oracle as orc
    probs, _ = orc.train_backwords(golden_lines("train_char.txt"), "", "\x00", "\x03", 0, 1, 10, 256)
    cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_BACKOFF, start_units=["\x00"])
    keys = cm.sample(n_global, 77)["ml2p"]
    srt, pos = c_oracle.rank_build(keys)
    p, c = c_oracle.rank_query(srt, pos, np.linspace(0.0, 60.0, 501))
    want = np.where(c > 0, p, 1.0)
    assert np.array_equal(cnt, c.astype(np.int64))
    assert cnt[0] == 0 and est[0] == 1.0                       # below every sample: the reference's `1`
    assert np.all(np.abs(est - want) <= 1e-9 * want)


def _merge_worker(rank, world, port, n_global, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import random
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from montecarlo_amd import dist as mcd
    from oracle import c_oracle, mc_oracle as orc
    random.seed(1000 + rank)                                   # ranks start apart ...
    mcd.use_group(True)
    drawn = random.getrandbits(64)                             # ... and agree after use_group
    assert mcd.active_group() is not None
    probs, _ = orc.train_backwords(golden_lines("train_char.txt"), "", "\x00", "\x03", 0, 1, 10, 256)
    cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_BACKOFF, start_units=["\x00"])
    first, count = mcd.shard_range(n_global, rank, world)
    res = cm.sample(count, 5, first_id=first, stride=40)
    local = {}
    for x, pwd in zip(res["ml2p"].tolist(), cm.strings(res)):
        local.setdefault(pwd, [x, 0])[1] += 1
    merged = mcd.merge_distinct([(p, x, c) for p, (x, c) in local.items()], mcd.active_group())
    with open(os.path.join(out_dir, f"merged{rank}.json"), "w") as f:
        import json
        json.dump({"drawn": drawn, "merged": merged}, f)
    mcd.use_group(None)
    assert mcd.active_group() is None
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sampled_pwds_merge_over_ranks(tmp_path):
    """`sample(size, sampled_pwds)` over shards: every rank ends with the dict a single process builds
    (lib4mc/MonteCarloParent.py:39-45) -- same strings, first-sampled order, counts -- and with the
    same `random` state (the drivers draw the Philox seed and random.sample from it)."""
    import json
    n_global, world = 4001, 3
    mp.spawn(_merge_worker, args=(world, _free_port(), n_global, str(tmp_path)), nprocs=world, join=True)
    got = [json.load(open(tmp_path / f"merged{r}.json")) for r in range(world)]
    assert got[0] == got[1] == got[2]
    from oracle import c_oracle, mc_oracle as orc
    probs, _ = orc.train_backwords(golden_lines("train_char.txt"), "", "\x00", "\x03", 0, 1, 10, 256)
    cm = c_oracle.COracleModel(orc.expand_table(probs), c_oracle.KIND_BACKOFF, start_units=["\x00"])
    res = cm.sample(n_global, 5, stride=40)
    want = {}
    for x, pwd in zip(res["ml2p"].tolist(), cm.strings(res)):
        want.setdefault(pwd, [x, 0])[1] += 1
    assert [tuple(r) for r in got[0]["merged"]] == [(p, x, c) for p, (x, c) in want.items()]
    assert sum(c for _, _, c in got[0]["merged"]) == n_global


def _poison_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from montecarlo_amd import dist as mcd
    pos = torch.tensor([0.0, 3.5, 7.25], dtype=torch.float64)
    if rank == 1:
        pos[1] = float("nan")                                  # what a failed rank build leaves in `positions`
    mcd.combine_positions(pos)
    try:
        mcd.estimates_from_positions(pos)
        verdict = "no error"
    except RuntimeError as exc:
        verdict = str(exc)
    with open(os.path.join(out_dir, f"verdict{rank}.txt"), "w") as f:
        f.write(verdict)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_nan_partial_from_one_shard_raises_on_every_rank(tmp_path):
    """A rank build that failed on one shard (NaN positions) must not turn into guess number 1 once the
    shards are summed: every rank raises."""
    world = 2
    mp.spawn(_poison_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert "NaN" in open(tmp_path / f"verdict{r}.txt").read()


def test_bench_dry_plan_gives_every_rank_and_step_its_own_sample_ids():
    """bench.py --gpus N --dry: the host-side plan of a multi-GPU run (no 8-GPU box is needed to check it): every
    (step, rank) draws its own Philox ids, a step's ranks tile n_global = N x samples ids without a gap."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry", "--steps", "5",
                          "--warmup", "2"], capture_output=True, text=True, timeout=300, check=True).stdout
    plan = json.loads(out.strip().splitlines()[-1])
    assert plan["dry"] and plan["n_gpus"] == 8 and plan["ranges_checked"] == 56
    assert plan["n_global_per_step"] == 8 * plan["samples_per_gpu_per_step"]
    assert plan["last"] == [6, 7, 55 * 10_000_000, 56 * 10_000_000]
    sys.path.insert(0, root)
    import bench
    assert [p[2:] for p in bench.id_plan(2, 2, 10)] == [(0, 10), (10, 20), (20, 30), (30, 40)]
