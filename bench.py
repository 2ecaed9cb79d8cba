#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json configs[1] -- backwords char-level backoff model
(threshold 10) trained on 100k synthetic Zipfian passwords; one "step" = the whole
guess-number hot path on one GPU:

    sample N=1e7 strings (k_sample) -> rank index: per distinct -log2 p its samples and the sum of 1/(n p) up to it
       (mc_rank_runs_build: all bisect_right ever reads; --rank-index sorted: the reference's two arrays, mc_rank_build)
    -> score T=1e6 test passwords (k_score) -> their stable order by -log2 p (mc_order_keys)
    -> bisect them into the curve (k_query)
    [-> all-reduce of the per-test partial sums when world_size > 1 (in the step's order; --overlap-comm puts
        it, the guess numbers and the copy on a stream of their own beside the next step's sampling)]
    -> monotone guess numbers (mc_guess_numbers) -> the guess numbers to pinned host memory (8 B per test
       password: three workgroups of the NEXT step's sampling launch move them while it samples, mc_sample_carry,
       and the last step's follow the loop inside the timed region, so K steps deliver K result sets;
       --d2h inline is the runtime's copy at the end of every step; --d2h-all adds the order and the sorted scores)

Scoring the test set and putting it in order do not depend on the samples: they run on a second stream beside
sampling and the rank build (--serial: one stream, in the order above), and the sampler launches 7 workgroups a
CU, which leaves every CU a slot for them (mc_sample_tune).

Inputs (compiled tables, encoded test set) are resident in HBM before the timed region.
Prints ONE JSON line on rank 0.  `python bench.py --gpus N --steps K --warmup W`; for N > 1
launch through torch.distributed.run (one rank per GPU, weak scaling: every GPU draws N).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("TQDM_DISABLE", "1")

from montecarlo_amd import synth  # noqa: E402
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo  # noqa: E402
from montecarlo_amd.engine import (DeviceRank, RunsBuffers, device_guess_numbers, device_order,  # noqa: E402
                                  device_order_begin, device_order_end, device_query_guess_numbers)
from montecarlo_amd import _native as nv  # noqa: E402
from montecarlo_amd import dist as mcd  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# HBM bytes per launch from the last committed PMC passes (profiles/hbm_traffic_latest.json: FETCH_SIZE doubled per
# the guide's gfx950 correction + WRITE_SIZE, separate rocprofv3 --pmc runs of this same command, measured by
# scripts/gpu_round.sh and published by scripts/publish_profile.py -- not inside this process).  The file names the
# commit and the digest of the native sources it was measured on: when the tree this process runs from has other
# sources, the numbers are stale and `traffic` is null.
TRAFFIC, TRAFFIC_SOURCE = {}, "none (profiles/hbm_traffic_latest.json absent)"
try:
    with open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")) as _f:
        _t = json.load(_f)
    from montecarlo_amd.build import sources_digest  # noqa: E402
    if _t.get("csrc_sha256") and _t["csrc_sha256"] == sources_digest():
        TRAFFIC = _t
        TRAFFIC_SOURCE = (f"{_t.get('source')} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of this command at commit "
                          f"{str(_t.get('commit'))[:12]}, native sources unchanged since)")
    else:
        TRAFFIC_SOURCE = (f"stale: {_t.get('source')} was measured at commit {str(_t.get('commit'))[:12]} on other "
                          f"native sources than this tree's")
except (OSError, ValueError):
    pass
# The same for k_sample's vector-ALU counters (profiles/valu_latest.json: scripts/valu_pass.sh, one rocprofv3 --pmc pass
# of this command and one of scripts/ubench/philox_mul as the yardstick; scripts/valu_summary.py says how they combine)
VALU = {"measured": None, "why": "profiles/valu_latest.json absent"}
try:
    with open(os.path.join(ROOT, "profiles", "valu_latest.json")) as _f:
        _v = json.load(_f)
    from montecarlo_amd.build import sources_digest  # noqa: E402
    if _v.get("csrc_sha256") and _v["csrc_sha256"] == sources_digest():
        _k = _v["k_sample_alone"]
        VALU = {"measured": _v.get("source"), "commit": str(_v.get("commit"))[:12],
                "busy": _k.get("valu_busy_by_mix"), "busy_bounds": [_k["valu_busy_vs_no_multiplies"], _k["valu_busy"]],
                "lanes_active": _k["lanes_active"], "wave_instructions_per_launch": _k["SQ_INSTS_VALU"],
                "scalar_instructions_per_launch": _k["SQ_INSTS_SALU"],
                "instructions_per_loop_iteration": {"valu": _k.get("valu_instructions_per_iteration"),
                                                    "salu": _k.get("salu_instructions_per_iteration")},
                "cycles_per_instruction": _v.get("cycles_per_instruction"),
                "what": "k_sample alone (the launches of roofline.alone); busy = the SIMDs' vector-ALU time over the "
                        "kernel's, pricing its own mix of v_mad_u64_u32 and other instructions at what "
                        "scripts/ubench/philox_mul costs per instruction at 8 waves a SIMD in the same --pmc pass; "
                        "bounds: against that loop without / with its multiplies"}
    else:
        VALU = {"measured": None, "why": f"stale: {_v.get('source')} was measured at commit {str(_v.get('commit'))[:12]} "
                                         f"on other native sources than this tree's"}
except (OSError, ValueError, KeyError):
    pass
# algorithmic bytes of k_sample: one 16-B draw-table record per draw, one more 16-B
# record for a draw whose bucket straddles a threshold, 8 B per threshold read from RCUM (counted
# from the compare counter: an upper bound, single-threshold buckets compare without a read),
# 8 B key written per sample.  All counts come from the device counters of the timed launches.
B_DRAW, B_INDIRECT, B_PROBE, B_SAMPLE = 16, 16, 8, 8
# rank build.  Run-length index (the default): ONE read of the keys (8 B a key) + 24 B written per
# distinct value.  Expanded arrays (--rank-index sorted): every key read once and written once in order (16 B), weight
# sums read it again (8 B), positions read it and write the running sum (16 B)
B_RANK_RUNS_KEY, B_RANK_RUNS_VALUE, B_RANK = 8, 24, 40


class _NoMark:
    """An event nobody reads: recording it is skipped (bench.py --time-every)."""
    @staticmethod
    def record(stream=None):
        pass


def C_vp(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr())


def C_stream(dev):
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def first_id(it: int, world: int, rank: int, n: int) -> int:
    """First Philox sample id of `rank` in step `it`: step-major, rank-minor, n ids each."""
    return (it * world + rank) * n


def id_plan(world: int, steps: int, n: int):
    """[(step, rank, first id, end id)] of a run, after checking that no two (step, rank) share a sample id, that a
    step's ranks tile one contiguous block of world * n ids (what a single GPU drawing n_global = world * n samples
    would use) and that nothing exceeds the 64-bit counter."""
    plan = [(it, r, first_id(it, world, r, n), first_id(it, world, r, n) + n) for it in range(steps) for r in range(world)]
    by_start = sorted(plan, key=lambda p: p[2])
    for a, b in zip(by_start, by_start[1:]):
        assert a[3] <= b[2], f"sample ids overlap: step {a[0]} rank {a[1]} and step {b[0]} rank {b[1]}"
    for it in range(steps):
        mine = [p for p in plan if p[0] == it]
        assert mine[0][2] == it * world * n and mine[-1][3] == (it + 1) * world * n
        assert all(x[3] == y[2] for x, y in zip(mine, mine[1:])), f"step {it}: the ranks' ranges leave a gap"
    assert by_start[-1][3] < 1 << 64
    return plan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (2.5 s of steps by default: a 20-step run is decided by one stall of a few ms, and the first few hundred steps
    # on a fresh box run 0.1-0.2 ms slower than the steady state -- 1.33-1.46 ms over 200 steps, 1.21 over 5000)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--samples", type=int, default=10_000_000, help="samples per GPU per step")
    ap.add_argument("--tests", type=int, default=1_000_000)
    ap.add_argument("--train-lines", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rank-index", choices=("runs", "sorted"), default="runs",
                    help="`runs` (default): the run-length rank index (mc_rank_runs_build: value, samples and summed "
                         "weights per distinct -log2 p; one read of the keys); `sorted`: the reference's sorted keys "
                         "and positions in full (mc_rank_build)")
    ap.add_argument("--second-stream", action="store_true", help=argparse.SUPPRESS)  # (the default)
    ap.add_argument("--serial", action="store_true",
                    help="everything on one stream in the step's own order (0.98 ms a step). Default: the test set is "
                         "scored and put in order -- neither depends on the samples -- on a second stream beside sampling "
                         "and the rank build, and the sampler runs 7 workgroups a CU instead of 14, which leaves every "
                         "CU a slot for those small launches (mc_sample_tune): 0.76 ms a step")
    ap.add_argument("--sample-wg-per-cu", type=int, default=0,
                    help="workgroups a CU of the sampling launch (default: 7 with the second stream, the library's 14 "
                         "with --serial)")
    ap.add_argument("--overlap-comm", action="store_true",
                    help="world > 1: all-reduce, guess numbers and copy on a stream of their own beside the next "
                         "step's sampling (the default keeps them in the step's own order)")
    ap.add_argument("--d2h", choices=("carry", "overlap", "inline"), default="carry",
                    help="copy of the guess numbers to the host: `carry` (default): the NEXT step's sampling launch takes "
                         "the copy along with three workgroups of its own (mc_sample_carry; the last step's copy "
                         "follows the loop, inside the timed region): 0.98 ms a step; `inline`: the runtime's copy on the "
                         "main stream inside the step (1.10 ms); `overlap`: on a copy stream beside the next step "
                         "(slower still). With --overlap-comm `carry` means `inline`")
    ap.add_argument("--d2h-all", action="store_true",
                    help="copy the test set's order and sorted scores (12 B per password, the same every step) to the "
                         "host every step as well, not only the guess numbers")
    ap.add_argument("--dry", action="store_true",
                    help="no GPU: check the sample-id plan of a --gpus N run (every rank and step draws its own "
                         "range of Philox ids, n_global = N x samples) and print it")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of the multi-rank control flow on a 1-GPU box: every rank uses cuda:0 and the "
                         "collective runs over gloo (the printed numbers mean nothing)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the legs after the timed region that run BASELINE configs 3, 4, 5 and the sub-word "
                         "scorer once at their per-GPU sizes (`other_configs` in the JSON line; about a minute)")
    ap.add_argument("--whole-order", action="store_true",
                    help="the main stream waits for the whole stable order of the test set (mc_order_keys) before its "
                         "queries; by default it waits for the sorted scores only (mc_order_keys_begin) and the "
                         "passwords' order is finished beside the queries (mc_order_keys_end)")
    ap.add_argument("--order-wait", action="store_true",
                    help="a step's sampling waits until the previous step's order of the passwords is whole (the second "
                         "stream is in order anyway and the sets are double-buffered: the wait is a choice, not a need; "
                         "0.595 ms a step with it, 0.591 without)")
    ap.add_argument("--no-fused-query", action="store_true",
                    help="one GPU, run index: answer the queries in a launch of their own (mc_rank_runs_query + "
                         "mc_guess_numbers, three launches) instead of inside the guess numbers' first kernel")
    ap.add_argument("--time-every", type=int, default=0,
                    help="record the stage events (stage_ms, the roofline's kernel_ms) in every K-th timed step only: "
                         "each recorded event is a host call and a barrier packet on the queue (nine of them a step "
                         "cost 4 us of its 0.64 ms).  Default: every 8th step from 64 timed steps on, else every step")
    ap.add_argument("--dump-last", default="",
                    help="rank 0 writes what the host received from the last timed step (guess numbers, how many of "
                         "them fit int64, order, sorted scores) to this .npz (tests/test_gpu_bench_modes.py)")
    ap.add_argument("--cpu-samples", type=int, default=0,
                    help="samples of the CPU baseline leg (default: the whole workload, about 10-20 s of one core)")
    args = ap.parse_args()

    if args.dry:
        plan = id_plan(args.gpus, args.warmup + args.steps, args.samples)
        print(json.dumps({"dry": True, "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                          "samples_per_gpu_per_step": args.samples, "n_global_per_step": args.samples * args.gpus,
                          "ranges_checked": len(plan), "first": plan[0], "last": plan[-1]}))
        return
    if os.environ.get("BENCH_TRACE_AFTER"):                 # debugging aid: where is every thread after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_TRACE_AFTER"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    plan = id_plan(world, args.warmup + args.steps, args.samples)       # (host asserts: disjoint ids, n_global)
    assert all(p[2] == first_id(p[0], world, p[1], args.samples) for p in plan if p[1] == rank)

    # ---------------- model + inputs (untimed) ----------------
    N, T = args.samples, args.tests
    train = synth.zipf_corpus(args.train_lines)
    model = BackWordsMonteCarlo(synth.as_file(train), splitter="", threshold=10)
    flat, dm = model._flat, model._dev
    test = synth.zipf_corpus(T, oov_frac=0.01, stream=1)
    syms, off = flat.encode_chars(test)
    d_syms = torch.from_numpy(syms.view(np.int32)).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    keys = torch.empty(N, dtype=torch.float64, device=dev)
    keys_sorted = torch.empty_like(keys)
    positions = torch.empty_like(keys)
    ws = torch.empty(nv.lib.mc_rank_workspace_bytes(N), dtype=torch.uint8, device=dev)
    runs_buf = RunsBuffers(N, dev) if args.rank_index == "runs" else None
    t_ml2p = torch.empty(T, dtype=torch.float64, device=dev)
    # two sets of result buffers, a stream of its own for the collective and one for the copies to the host:
    # a step's all-reduce (world > 1) runs beside the next step's SAMPLING and the main stream waits for it
    # before the next rank build; the copies run beside whatever follows and the main stream waits for them only
    # when it is about to write the same set again, two steps later
    q_pos = [torch.empty(T, dtype=torch.float64, device=dev) for _ in range(2)]
    t_sorted = [torch.empty(T, dtype=torch.float64, device=dev) for _ in range(2)]
    t_vals = [torch.empty(T, dtype=torch.float64, device=dev) for _ in range(2)]     # (mc_order_keys_begin: the scores ascending, as numbers)
    t_order = [torch.empty(T, dtype=torch.int32, device=dev) for _ in range(2)]
    # (the guess numbers and the count of those that are exact share one buffer: ONE copy to the host a step)
    # (T + 1 words rounded up to whole 16 bytes: mc_sample_carry moves 16-byte words)
    TB = (T + 2) & ~1
    g_both = [torch.zeros(TB, dtype=torch.int64, device=dev) for _ in range(2)]
    g_rank = [b[:T] for b in g_both]
    g_exact = [b[T:T + 1] for b in g_both]
    h_both = [torch.zeros(TB, dtype=torch.int64).pin_memory() for _ in range(2)]
    host = [{"rank": h_both[i][:T], "order": torch.empty(T, dtype=torch.int32).pin_memory(),
             "ml2p": torch.empty(T, dtype=torch.float64).pin_memory(), "exact": h_both[i][T:T + 1]}
            for i in range(2)]
    ws_order = torch.empty(nv.lib.mc_order_workspace_bytes(T), dtype=torch.uint8, device=dev)
    ws_gn = torch.empty(nv.lib.mc_guess_numbers_workspace_bytes(T), dtype=torch.uint8, device=dev)
    reduced, copied, copied_on = [None, None], [None, None], [None, None]
    order_done = [None, None]                               # (second stream: when a set's order of the passwords is whole)
    comm = torch.cuda.Stream(device=dev) if world > 1 and args.overlap_comm else None
    copy = torch.cuda.Stream(device=dev)
    q_cnt = torch.empty(T, dtype=torch.int64, device=dev)
    seed = 20250304

    ev = {k: [] for k in ("sample", "rank", "score", "order", "query", "guess_numbers", "d2h")}
    side = None if args.serial else torch.cuda.Stream(device=dev)
    if side is not None or args.sample_wg_per_cu:
        # one generation of sampling workgroups (round 2, 63 registers a lane: 14 a CU 0.807 ms a step, 12 0.787, 10 0.796,
        # 8 0.812, 7 0.759, 6 0.831; with the lean kernel's 56 registers, late round 3: 6 0.675, 7 0.626, 8 0.619, 9 and 10 below)
        dm.tune(wg_per_cu=args.sample_wg_per_cu or int(os.environ.get("MC_SAMPLE_WG_PER_CU", "8")),
                skew=float(os.environ.get("MC_SAMPLE_SKEW", "0.65")))  # (ranges' skew beside the second stream: 0.55 0.620 ms a step, 0.65 0.613, 0.75 0.616, 0.85 0.630)
    main_stream = torch.cuda.current_stream(dev)

    rk_last = [None]
    carry_mode = args.d2h == "carry" and comm is None
    pending = [None]                                        # carry mode: the result set the next sampling launch takes along

    def flush_carry():
        if pending[0] is not None:                          # (the last step's guess numbers: the runtime's copy)
            h_both[pending[0]].copy_(g_both[pending[0]], non_blocking=True)
            pending[0] = None

    def step(it: int, timed: bool):
        """sample -> rank build -> [score -> order] -> queries -> guess numbers; the bracketed part runs on a
        second stream from the start of the step unless --serial; the guess numbers reach the host with the
        next step's sampling launch (--d2h)."""
        if timed:
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
        else:                                                # (only what orders the streams)
            needed = (0, 6, 12) if carry_mode else (0, 6, 8, 11, 12)      # (what orders the streams and the copies)
            marks = [torch.cuda.Event() if i in needed else _NoMark for i in range(13)]
        slot = it & 1
        first = first_id(it, world, rank, N)                # disjoint Philox sample ids per rank and step
        if side is not None and not args.whole_order and order_done[slot ^ 1] is not None and args.order_wait:
            # --order-wait: a step's results are whole before the next one samples, and the order's last launches do not
            # sit beside the sampler (0.619 ms a step against 0.622 when the second stream was the longer one by far; since
            # its launches were trimmed, r03_o, the wait costs: 0.595 against 0.591).  Without it nothing is unordered:
            # the second stream runs its launches in order and every set of buffers is written two steps apart.
            main_stream.wait_event(order_done[slot ^ 1])
            order_done[slot ^ 1] = None
        marks[0].record()
        if side is not None:
            # the test set's scores and their order do not depend on the samples: a second stream computes them
            # beside sampling and the rank build (small, latency-bound launches under two chip-wide ones)
            side.wait_event(marks[0])
            with torch.cuda.stream(side):
                if copied[slot] is not None:                # this set's copies of two steps ago
                    side.wait_event(copied[slot])
                    copied[slot] = None
                marks[3].record()
                dm.score_device(d_syms, d_off, T, out=t_ml2p)
                marks[4].record()
                if args.whole_order:
                    device_order(t_ml2p, out_sorted=t_sorted[slot], out_order=t_order[slot], workspace=ws_order)
                    marks[6].record()
                    marks[12] = marks[6]
                else:
                    # the queries need the scores in ascending order, not which password has which: the main stream
                    # waits for the key sort (marks[6]) and the passwords' order is finished beside its queries
                    device_order_begin(t_ml2p, t_sorted[slot], t_vals[slot], ws_order)
                    marks[6].record()
                    device_order_end(t_ml2p, t_sorted[slot], t_order[slot], ws_order)
                    marks[12].record()
                order_done[slot] = marks[12]
        if carry_mode and pending[0] is not None:
            dm.sample(N, seed, first_id=first, out=keys, carry=(g_both[pending[0]], h_both[pending[0]]))
            pending[0] = None
        else:
            dm.sample(N, seed, first_id=first, out=keys)
        marks[1].record()
        for i, done in enumerate(reduced):                  # the previous step's all-reduce ran beside k_sample
            if done is not None:
                main_stream.wait_event(done)
                reduced[i] = None
        if runs_buf is not None:
            rk = DeviceRank(keys, n_global=N * world, runs=runs_buf)
        else:
            rk = DeviceRank(keys, n_global=N * world, workspace=ws, keys_sorted=keys_sorted, positions=positions)
        rk_last[0] = rk
        marks[2].record()
        if side is None:
            if copied[slot] is not None:
                if copied_on[slot] is not main_stream:      # (a copy on this stream is behind us already)
                    main_stream.wait_event(copied[slot])
                copied[slot] = None
            marks[3] = marks[2]                             # (one stream: a stage starts where the last one ended;
            dm.score_device(d_syms, d_off, T, out=t_ml2p)   #  every recorded event is a barrier packet on the queue)
            marks[4].record()
            device_order(t_ml2p, out_sorted=t_sorted[slot], out_order=t_order[slot], workspace=ws_order)
            marks[6].record()
            marks[5] = marks[6]
            marks[12] = marks[6]
        else:
            main_stream.wait_event(marks[6])
            marks[5].record()
        fused = world == 1 and rk.runs is not None and not args.no_fused_query
        t_query = t_vals[slot] if (side is not None and not args.whole_order) else t_sorted[slot]
        if fused:
            # one GPU, run index: the bisection inside the guess numbers' first kernel (mc_rank_runs_guess_numbers:
            # two launches where query + guess numbers are three); timed as one stage, under `guess_numbers`
            marks[7] = marks[5]
            device_query_guess_numbers(rk, t_query, True, out_pos=q_pos[slot], out_cnt=q_cnt, out=g_rank[slot],
                                       workspace=ws_gn, exact=g_exact[slot])
            marks[8].record()
            ready = marks[8]
        else:
            rk.query(t_query, out_pos=q_pos[slot], out_cnt=q_cnt)
            marks[7].record()
        if fused:
            pass
        elif world == 1:
            device_guess_numbers(q_pos[slot], q_cnt, True, out=g_rank[slot], workspace=ws_gn, exact=g_exact[slot])
            marks[8].record()
            ready = marks[8]
        elif comm is None:
            # ONE RCCL all-reduce (8 MB) per step, in the step's own order, then the guess numbers from the summed
            # positions (a summed position is > 0 exactly when some shard counted a sample).  Not beside the next
            # step's sampling (--overlap-comm does that): on one GPU every kernel put beside the sampler on another
            # stream cost more than it hid (the 8 MB copy alone: 1.52 against 1.33 ms a step).
            if args.rehearse_on_one_gpu:
                torch.cuda.synchronize()                    # (gloo on a shared GPU: see below)
            mcd.combine_positions(q_pos[slot])
            marks[9].record()
            device_guess_numbers(q_pos[slot], None, True, out=g_rank[slot], workspace=ws_gn, exact=g_exact[slot])
            marks[8].record()
            ready = marks[8]
        else:
            # ONE RCCL all-reduce (8 MB) per step, on the collective's stream behind the queries, then the guess
            # numbers from the summed positions (a summed position is > 0 exactly when some shard counted a sample)
            asked = torch.cuda.Event()
            asked.record(main_stream)
            comm.wait_event(asked)
            if args.rehearse_on_one_gpu:
                # gloo stages GPU tensors through the host on a stream of its own; entered with this
                # process's kernels still queued while the other rank computes on the same GPU it was
                # seen to stall (both ranks inside all_reduce, neither finishing).  Idle the GPU first.
                torch.cuda.synchronize()
            with torch.cuda.stream(comm):
                mcd.combine_positions(q_pos[slot])
                marks[9].record()
                device_guess_numbers(q_pos[slot], None, True, out=g_rank[slot], workspace=ws_gn, exact=g_exact[slot])
                marks[8].record()
                reduced[slot] = marks[8]
            ready = marks[8]
        lane = copy if args.d2h == "overlap" else (comm if comm is not None else main_stream)
        if carry_mode:
            pending[0] = slot                               # (travels with the next step's sampling launch)
            if args.d2h_all:
                if side is not None and order_done[slot] is not None:
                    main_stream.wait_event(order_done[slot])
                host[slot]["order"].copy_(t_order[slot], non_blocking=True)
                host[slot]["ml2p"].copy_(t_sorted[slot], non_blocking=True)
            if timed:
                for k, (a_, b_) in (("sample", (0, 1)), ("rank", (1, 2)), ("score", (3, 4)), ("order", (4, 12)),
                                    ("query", (5, 7))):
                    ev[k].append((marks[a_], marks[b_]))
                ev["guess_numbers"].append((marks[7] if world == 1 else marks[9], marks[8]))
                ev["d2h"].append((ready, ready))
            return
        if lane is main_stream or lane is comm:
            marks[10] = ready                               # (same stream as the guess numbers: nothing to wait for)
        else:
            lane.wait_event(ready)
        with torch.cuda.stream(lane):
            if marks[10] is not ready:
                marks[10].record()
            # the guess numbers travel every step (8 B per test password: they are what the samples decide); the
            # order of the test set and its sorted scores do not depend on the samples -- a caller fetches them
            # once per test set, here after the timed region for the checks (--d2h-all copies them every step)
            h_both[slot].copy_(g_both[slot], non_blocking=True)
            if args.d2h_all:
                if side is not None and order_done[slot] is not None:
                    lane.wait_event(order_done[slot])
                host[slot]["order"].copy_(t_order[slot], non_blocking=True)
                host[slot]["ml2p"].copy_(t_sorted[slot], non_blocking=True)
            marks[11].record()
            copied[slot], copied_on[slot] = marks[11], lane
        if timed:
            ev["sample"].append((marks[0], marks[1]))
            ev["rank"].append((marks[1], marks[2]))
            ev["score"].append((marks[3], marks[4]))
            ev["order"].append((marks[4], marks[12]))
            ev["query"].append((marks[5], marks[7]))
            ev["guess_numbers"].append((marks[7] if world == 1 else marks[9], marks[8]))
            ev["d2h"].append((marks[10], marks[11]))

    def fence():
        torch.cuda.synchronize()                            # (every stream of the device: the collective's too)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    time_every = args.time_every if args.time_every > 0 else (8 if args.steps >= 64 else 1)
    first_step_ms = None
    for w in range(args.warmup):
        if w == 0:                                          # the very first step of the process, alone and waited for
            torch.cuda.synchronize()
            t_first = time.perf_counter()
        step(w, False)
        if w == 0:
            torch.cuda.synchronize()
            first_step_ms = (time.perf_counter() - t_first) * 1e3
    torch.cuda.synchronize()
    dm.check()                                              # clears counters; raises on a diverged sample
    fence()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(args.warmup + s, s % time_every == 0)
    flush_carry()
    issued = time.perf_counter() - t0                       # the host's share: launching, not waiting
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    cnt = dm.check()
    ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
    if carry_mode:
        ms["d2h"] = None                                    # (carried in the sampling launch: inside `sample`)

    # sanity on the last step (not timed): finite keys, sorted output (run index: the arrays written from the runs)
    sort_info = rk_last[0].check()
    if runs_buf is not None:
        keys_sorted, positions = rk_last[0].keys_sorted, rk_last[0].positions
        D = sort_info["distinct_values"]
        assert sort_info["keys_counted"] == N and 0 < D <= N
        assert bool((runs_buf.vals[1:D] > runs_buf.vals[:D - 1]).all()) and int(runs_buf.cum_count[D - 1]) == N
    assert bool(torch.isfinite(keys).all()) and bool((keys_sorted[1:] >= keys_sorted[:-1]).all())
    assert bool(torch.isfinite(positions).all()) and bool((positions[1:] >= positions[:-1]).all())

    # the last two steps' results as the host received them: guess numbers strictly increasing as far as they
    # fit int64 (beyond, the array carries the estimates for the caller's big integers: passwords the model
    # cannot spell sit at the sum of all weights), the order a permutation under which the scores ascend
    if not args.d2h_all:
        for sl in range(min(2, args.steps + args.warmup)):
            host[sl]["order"].copy_(t_order[sl])
            host[sl]["ml2p"].copy_(t_sorted[sl])
    for h in host[:min(2, args.steps + args.warmup)]:
        exact = int(h["exact"][0])
        assert 0 < exact <= T and int(h["order"][0]) != -1
        r = h["rank"].numpy()
        assert r[0] >= 1 and bool((np.diff(r[:exact]) >= 1).all())
        est = r[exact:].view(np.float64)
        assert bool((est >= 2.0 ** 62).all()) and bool((np.diff(est) >= 0).all())
        assert bool((np.diff(h["ml2p"].numpy()) >= 0).all())
    last = host[(args.warmup + args.steps - 1) & 1]
    assert np.array_equal(np.sort(last["order"].numpy().view(np.uint32)), np.arange(T, dtype=np.uint32))
    assert np.array_equal(t_ml2p.cpu().numpy()[last["order"].numpy().view(np.uint32)], last["ml2p"].numpy())

    if args.dump_last and rank == 0:
        np.savez(args.dump_last, rank=last["rank"].numpy(), exact=last["exact"].numpy(), order=last["order"].numpy(),
                 ml2p=last["ml2p"].numpy())

    # ---- what the host received from THIS arrangement of streams against a serial recompute and the C oracle ----
    arrangement = check_arrangement(args, dm, model, flat, host, t_ml2p, d_syms, d_off, N, T, world, rank, seed, dev) \
        if world == 1 else None

    # ---- one cold step: no hint from earlier steps in any workspace, alone on the GPU, waited for ----
    cold_step_ms = None
    if world == 1:
        if runs_buf is not None:
            nv.check(nv.lib.mc_rank_runs_forget(C_vp(runs_buf.ws), N, C_stream(dev)))
        else:
            nv.check(nv.lib.mc_workspace_forget(C_vp(ws), N, C_stream(dev)))
        nv.check(nv.lib.mc_workspace_forget(C_vp(ws_order), T, C_stream(dev)))
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        step(args.warmup + args.steps, False)
        flush_carry()
        torch.cuda.synchronize()
        cold_step_ms = (time.perf_counter() - t_c) * 1e3

    # the sampling launch with the GPU to itself (the library's own geometry, nothing carried, no second stream):
    # 50 launches after the timed region, for the roofline line beside the in-step figure
    dm.tune()
    for _ in range(5):
        dm.sample(N, seed, first_id=0, out=keys)
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for i in range(50):
        dm.sample(N, seed, first_id=i * N, out=keys)
    a1.record()
    torch.cuda.synchronize()
    alone_ms = a0.elapsed_time(a1) / 50
    dm.check()

    if rank == 0:
        draws, probes = cnt["draws"] / args.steps, cnt["cdf_probes"] / args.steps
        indirect = cnt["indirect_draws"] / args.steps
        bytes_per_launch = draws * B_DRAW + indirect * B_INDIRECT + probes * B_PROBE + N * B_SAMPLE
        if carry_mode:                                      # the launch also reads and writes the previous result set
            bytes_per_launch += 2 * 8 * TB
        achieved = bytes_per_launch / (ms["sample"] * 1e-3) / 1e9
        rank_bytes = (N * B_RANK_RUNS_KEY + sort_info["distinct_values"] * B_RANK_RUNS_VALUE) if runs_buf is not None \
            else N * B_RANK
        gn_ms = ms["score"] + ms["order"] + ms["query"] + ms["guess_numbers"]
        rank_achieved = rank_bytes / (ms["rank"] * 1e-3) / 1e9
        out = {
            "metric": "sampled passwords/sec (node) + guess numbers/sec, 4-gram backoff, 1/2/4/8 GPU",
            "value": N * world * args.steps / dt,
            "unit": "sampled passwords/s (through sample+sort+scan+score+rank)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: backwords char-level backoff (threshold 10, 100k synthetic Zipf "
                                   "training lines), 1e7 samples + 1e6 test pwds per GPU per step",
                       "rank_index": ("csrc/mc_runs.hip: run-length index (value, samples, summed weights per distinct "
                                      "-log2 p; heavy values counted in LDS, the rest ordered by value range batch by "
                                      "batch and counted per range; no workgroup waits for another)"
                                      if runs_buf is not None else
                                      "csrc/mc_sort.hip: sorted keys + positions in full (heavy values counted, MSD radix "
                                      "on the rest, in-LDS finish; no workgroup waits for another)"),
                       "samples_per_gpu": N, "test_pwds": T, "contexts": int(flat.info[nv.INFO_NCTX]),
                       "entries": int(flat.info[nv.INFO_NENT]), "table_bytes": int(flat.info[nv.INFO_DEVICE_BYTES])},
            "stage_ms": ms, "test_set_on_second_stream": side is not None,
            # (one GPU + run index: the queries are answered inside the guess numbers' first kernel; stage `query` is 0
            # and `guess_numbers` holds both)
            "query_inside_guess_numbers": world == 1 and runs_buf is not None and not args.no_fused_query,
            "queries_wait_for": ("the whole order of the test set (mc_order_keys)" if (args.serial or args.whole_order)
                                 else "the sorted scores (mc_order_keys_begin); the passwords' order is finished beside them "
                                      "(mc_order_keys_end, inside the step: `order` in stage_ms spans both)"),
            # one step alone and waited for: the first of the process (code objects load, allocator warms up) and one
            # after the timed region with every workspace's hint forgotten (what ONE estimate costs a caller)
            "first_step_ms": first_step_ms, "cold_step_ms": cold_step_ms,
            "arrangement_check": arrangement,
            "stage_ms_note": ("score and order run on a second stream beside sample and rank (their times include "
                              "waiting for free slots); the step is sample + rank + query + guess_numbers"
                              if side is not None else "one stream: the step is the sum of the stages"),
            "stage_events_every": time_every,
            "d2h": ("carry: each step's guess numbers travel with the next step's sampling launch (mc_sample_carry), "
                    "the last step's after the loop, inside the timed region" if carry_mode else args.d2h),
            "host_issue_ms_per_step": issued / args.steps * 1e3,
            "sampling_only_per_s": N * world / (ms["sample"] * 1e-3),
            # the step as round 1 defined it (sample + rank build + score + bisection, nothing ordered, ranked or
            # copied), from the same launches' stage times: the figure to hold against round 1's 9.6e9
            # (--serial only: on two streams the scoring's stage time is mostly waiting beside the sampler)
            "round1_step_samples_per_s": (N * world / ((ms["sample"] + ms["rank"] + ms["score"] + ms["query"]) * 1e-3)
                                          if side is None else None),
            # score + stable order + bisection + monotone guess numbers (device time of those four stages); the
            # copy of ranks/order/scores to the host (`d2h` in stage_ms) runs beside the next step
            # every step delivers T guess numbers to the host
            "guess_numbers_per_s": T * world * args.steps / dt,
            # (--serial only: T over the device time of score + order + query + guess numbers, one after the other)
            "guess_numbers_stages_per_s": T / (gn_ms * 1e-3) if side is None else None,
            "d2h_bytes_per_step": (20 if args.d2h_all else 8) * T + 8 * (TB - T), "guess_numbers_within_int64": exact / T,
            "draws_per_sample": draws / N, "indirect_draws_per_draw": indirect / max(draws, 1),
            "threshold_compares_per_draw": probes / max(draws, 1),
            # the longest single launch; HBM traffic from profiles/*pmc_hbm_traffic.json (rocprofv3 --pmc)
            # ("hbm" is the roofline the contract prices the kernel against; by its counters the kernel is bound by
            # vector-ALU issue and the latency of its two dependent gathers an iteration: `valu`)
            "roofline": {"bound": "hbm", "bound_by_counters": "vector-ALU issue + gather latency (see valu; DESIGN.md 3.1)",
                         "kernel": "k_sample", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": TRAFFIC.get("k_sample"),
                         "traffic_source": TRAFFIC_SOURCE, "valu": VALU,
                         "bytes_per_launch": bytes_per_launch, "kernel_ms": ms["sample"],
                         "note": "timed inside the step: the copy workgroups of mc_sample_carry and the second "
                                 "stream's launches share the GPU with it" if side is not None else "",
                         "alone": {"kernel_ms": alone_ms, "achieved": (bytes_per_launch - (2 * 8 * TB if carry_mode else 0))
                                   / (alone_ms * 1e-3) / 1e9,
                                   "frac": (bytes_per_launch - (2 * 8 * TB if carry_mode else 0)) / (alone_ms * 1e-3) / 1e9
                                   / HBM_PEAK_GBS,
                                   "what": "50 launches of the same 1e7 samples after the timed region, GPU to itself, "
                                           "the library's default geometry (8 workgroups a CU up to 2^24 samples a launch, 16 beyond), nothing carried"}},
            # the rank build as a whole, timed together: the run index against ONE read of the keys + 24 B per
            # distinct value; the expanded arrays against the 40 B/key a sort-and-scan cannot do without
            "roofline_rank_build": {"bound": "hbm", "kernel": "mc_rank_runs_build (k_rl_sniff, k_rl_strip, k_rl_bins, "
                                    "k_rl_final)" if runs_buf is not None else "mc_rank_build (k_sniff, k_strip, "
                                    "k_scatter, k_count, k_deep, k_finish, k_fill, k_weight_sums, k_positions)",
                                    "achieved": rank_achieved,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rank_achieved / HBM_PEAK_GBS,
                                    "traffic": TRAFFIC.get("mc_rank_build"), "bytes_per_launch": rank_bytes,
                                    "kernel_ms": ms["rank"], "sort": sort_info},
        }
        if not args.no_other_configs and world == 1 and N >= 1_000_000:
            del keys, keys_sorted, positions, ws                   # (the legs below bring their own buffers)
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, test, args.cpu_samples or N)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def check_arrangement(args, dm, model, flat, host, t_ml2p, d_syms, d_off, N, T, world, rank, seed, dev):
    """The guess numbers the host received from the LAST timed step -- through this run's arrangement of streams, the
    carried copy and the reused workspaces -- against (a) the same step computed again on one stream with fresh
    buffers from the same sample ids: order and sorted scores equal bit for bit, guess numbers equal (bit for bit with
    --rank-index sorted; with the run index, whose sums follow the build's own bins, within 1e-9 -- in practice one
    ulp of the float64 position -- and at most 2 of those below 2^50 off by one where ceil() sits on an integer); (b) on a cut of 2e5
    samples and 3e4 test passwords, the C oracle's ranks (reference lib4mc/MonteCarloLib.py:34-58): within 1e-9, and
    at most 2 guess numbers differing where ceil() sits on an exact integer."""
    from oracle import c_oracle
    it = args.warmup + args.steps - 1
    got = host[it & 1]
    first = first_id(it, world, rank, N)
    keys = dm.sample(N, seed, first_id=first)["ml2p"]
    rk = DeviceRank(keys, n_global=N * world, index=args.rank_index)
    sc = dm.score_device(d_syms, d_off, T)
    srt, order = device_order(sc)
    pos, cnt = rk.query(srt)
    ranks, exact, _ = device_guess_numbers(pos, cnt, True)
    torch.cuda.synchronize()
    # order and sorted scores do not depend on the samples: bit for bit.  Guess numbers: with the expanded arrays bit
    # for bit; the run index sums a value range's weights in the order of ITS bins, which follow the sample a build
    # (or the build whose hint it took) looked at: positions agree to ~1e-15 and a guess number may differ where
    # ceil() sits on an exact integer
    same_order = bool(torch.equal(order.cpu(), got["order"]) and
                      torch.equal(srt.cpu().view(torch.int64), got["ml2p"].view(torch.int64)))
    k = int(exact.item())
    a, b = ranks.cpu().numpy()[:k], got["rank"].numpy()[:k]
    differing = int(np.sum(a != b))
    # (a guess number beyond 2^50 is a rounded float64 position: one ulp of the sum IS a different integer there)
    small = b < (1 << 50)
    differing_small = int(np.sum(a[small] != b[small]))
    rel_gn = float(np.max(np.abs(a - b) / np.maximum(b, 1))) if k else 0.0
    res = {"order_and_scores_bit_identical": same_order, "exact_count_equal": k == int(got["exact"][0]),
           "guess_numbers_compared": k, "guess_numbers_differing_from_serial_recompute": differing,
           "of_them_below_2^50": differing_small, "guess_numbers_max_rel_diff": rel_gn}
    assert same_order and res["exact_count_equal"], "the step's arrangement of streams changed what the host received"
    assert rel_gn <= 1e-9 and (differing_small <= 2 if args.rank_index == "runs" else differing == 0), res
    # the C oracle on a cut: its own samples (same Philox ids), rank build, scores, order, bisection, ranks
    n_cut, t_cut = min(N, 200_000), min(T, 30_000)
    cm = c_oracle.COracleModel(model.nwords, c_oracle.KIND_BACKOFF, start_units=["\x00"])
    want_keys = cm.sample(n_cut, seed, first_id=first)["ml2p"]
    assert np.array_equal(want_keys.view(np.uint64), keys[:n_cut].cpu().numpy().view(np.uint64))
    osrt, opos = c_oracle.rank_build(want_keys)
    sc_cut = sc[:t_cut].contiguous()
    o_order = np.argsort(sc_cut.cpu().numpy(), kind="stable")
    o_scores = sc_cut.cpu().numpy()[o_order]
    o_pos, o_cnt = c_oracle.rank_query(osrt, opos, o_scores)[:2]
    est = np.where(np.asarray(o_cnt) > 0, o_pos, 1.0)
    want, prev = [], 0
    for e in est.tolist():
        prev = int(np.ceil(max(e, prev + 1)))
        want.append(prev)
    rk2 = DeviceRank(keys[:n_cut].contiguous(), index=args.rank_index)
    srt2, order2 = device_order(sc_cut)
    pos2, cnt2 = rk2.query(srt2)
    ranks2, exact2, _ = device_guess_numbers(pos2, cnt2, True)
    torch.cuda.synchronize()
    assert np.array_equal(order2.cpu().numpy().view(np.uint32), o_order.astype(np.uint32))
    assert np.array_equal(cnt2.cpu().numpy(), np.asarray(o_cnt, dtype=np.int64))
    p2 = pos2.cpu().numpy()
    rel = float(np.max(np.abs(p2 - o_pos) / np.maximum(np.abs(o_pos), 1e-300)))
    assert rel <= 1e-9, rel
    k = int(exact2.item())
    g = ranks2.cpu().numpy()[:k]
    differ = int(np.sum(g != np.array(want[:k], dtype=np.int64)))
    assert differ <= 2 and all(abs(int(a) - b) <= 1e-9 * b + 1 for a, b in zip(g.tolist(), want[:k]))
    res.update({"oracle_cut": f"{n_cut} samples, {t_cut} test passwords", "positions_max_rel_diff_vs_oracle": rel,
                "guess_numbers_differing_from_oracle": differ, "guess_numbers_compared_with_oracle": k})
    return res


def other_configs(dev):
    """BASELINE configs 3, 4, 5 and the sub-word scorer once each at their per-GPU sizes, after the timed region:
    device time of the launches (HIP events; inputs, outputs and workspaces resident, nothing copied or allocated
    between the events), the bytes the kernels have to move by their own counters, and the fraction of 8 TB/s."""
    import tempfile

    from montecarlo_amd.bpe_simulator import BpePcfgSim
    from montecarlo_amd.nwords_simulator_v2 import NWords2MonteCarlo
    out = {}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    def sample_line(dm, n, ms, cnt, reps):
        draws, ind, probes = cnt["draws"] / reps, cnt["indirect_draws"] / reps, cnt["cdf_probes"] / reps
        byts = draws * B_DRAW + ind * B_INDIRECT + probes * B_PROBE + n * B_SAMPLE
        return {"ms": ms, "samples_per_s": n / (ms * 1e-3), "draws_per_sample": draws / n, "bytes": byts,
                "frac_of_hbm": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # ---- config 3: nwords_v2 6-gram, 1e10 samples over 8 GPUs = 1.25e9 a GPU: sample, rank build, 1e6 queries ----
    free, _ = torch.cuda.mem_get_info()
    n3 = 1_250_000_000 if free > 70 * 2 ** 30 else 125_000_000
    m3 = NWords2MonteCarlo(synth.as_file(synth.v2_corpus(100000)), 6)
    keys = torch.empty(n3, dtype=torch.float64, device=dev)
    srt, pos = torch.empty_like(keys), torch.empty_like(keys)
    ws = torch.empty(nv.lib.mc_rank_workspace_bytes(n3), dtype=torch.uint8, device=dev)
    nv.check(nv.lib.mc_workspace_forget(C_vp(ws), n3, C_stream(dev)))
    m3._dev.check()
    ms = timed(lambda: m3._dev.sample(n3, 5, first_id=3 * n3, out=keys), reps=2)
    c3 = sample_line(m3._dev, n3, ms, m3._dev.check(), 3)
    ms_rank = timed(lambda: DeviceRank(keys, n_global=8 * n3, workspace=ws, keys_sorted=srt, positions=pos), reps=2)
    rk = DeviceRank(keys, n_global=8 * n3, workspace=ws, keys_sorted=srt, positions=pos)
    x = srt[:: max(1, n3 // 1_000_000)][:1_000_000].contiguous()
    qp, qc = torch.empty_like(x), torch.empty(x.numel(), dtype=torch.int64, device=dev)
    ms_q = timed(lambda: rk.query(x, out_pos=qp, out_cnt=qc))
    info = rk.check()
    assert info["error"] == 0 and bool((qc[1:] >= qc[:-1]).all()) and int(qc[-1]) <= n3
    out["config3_nwords_v2_6gram"] = {
        "what": f"{n3} samples (one GPU's share of 1e10 over 8) of the 100k-line synthetic v2 6-gram model: k_sample, "
                f"mc_rank_build (expanded arrays: beyond the run index's 2^26 keys), {x.numel()} queries",
        "k_sample": c3,
        "mc_rank_build": {"ms": ms_rank, "keys_per_s": n3 / (ms_rank * 1e-3), "bytes": n3 * B_RANK,
                          "frac_of_hbm": n3 * B_RANK / (ms_rank * 1e-3) / 1e9 / HBM_PEAK_GBS, "sort": info},
        "k_query": {"ms": ms_q, "queries_per_s": x.numel() / (ms_q * 1e-3)},
        "pipeline_samples_per_s": n3 / ((ms + ms_rank + ms_q) * 1e-3)}
    del keys, srt, pos, ws, rk, x, qp, qc
    torch.cuda.empty_cache()

    # ---- config 4: BPE-PCFG, 1e9 samples over 8 GPUs = 1.25e8 a GPU; 1e6 test passwords scored ----
    n4 = 125_000_000
    with tempfile.TemporaryDirectory() as tmp:
        synth.zipf_bpe_dir(tmp, np.random.default_rng(4))
        m4 = BpePcfgSim(tmp)
    keys = torch.empty(n4, dtype=torch.float64, device=dev)
    m4._dev.check()
    ms = timed(lambda: m4._dev.sample(n4, 3, out=keys), reps=2)
    c4 = sample_line(m4._dev, n4, ms, m4._dev.check(), 3)
    res = m4._dev.sample(200_000, 5, want_len=True, sym_stride=m4._flat.max_segments)
    pw = list(dict.fromkeys(m4._flat.decode_entries(res["syms"].cpu().numpy(), res["len"].cpu().numpy())))
    pw = (pw * (1_000_000 // len(pw) + 1))[:1_000_000]
    cps, cls, off = m4.encode(pw)
    d_cps = torch.from_numpy(np.ascontiguousarray(cps, dtype=np.uint32).view(np.int32)).to(dev)
    d_cls = torch.from_numpy(np.ascontiguousarray(cls, dtype=np.uint8)).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.int64)).to(dev)
    sc = torch.empty(len(pw), dtype=torch.float64, device=dev)
    m4._dev.check()
    ms_s = timed(lambda: m4._dev.score_bpe_device(d_cps, d_cls, d_off, len(pw), sc))
    cs = m4._dev.check()
    sbytes = cs["score_probes"] / 4 * 16 + cps.nbytes + cls.nbytes + off.nbytes + sc.numel() * 8
    assert bool(torch.isfinite(sc).all())
    out["config4_bpe_pcfg"] = {
        "what": f"{n4} samples (one GPU's share of 1e9 over 8) of a synthetic BPE-PCFG model (400 structures, terminal "
                f"rows up to 1e5 entries): k_sample_bpe; {len(pw)} test passwords scored: k_score_bpe",
        "k_sample_bpe": c4,
        "k_score_bpe": {"ms": ms_s, "passwords_per_s": len(pw) / (ms_s * 1e-3), "bytes": sbytes,
                        "hash_probes_per_password": cs["score_probes"] / 4 / len(pw),
                        "frac_of_hbm": sbytes / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    del keys, d_cps, d_cls, d_off, sc
    torch.cuda.empty_cache()

    # ---- sub-word backoff model: calc_ml2p's DFS over segmentations on 1e6 resident passwords ----
    sub = synth.subword_corpus(100000)
    ms_model = BackWordsMonteCarlo(synth.as_file(sub), splitter="\t", threshold=10)
    test = ["".join(ln.split("\t")) for ln in synth.subword_corpus(1_000_000, stream=1)]
    cps, off, max_len = ms_model._flat.encode_text(test)
    d_cps = torch.from_numpy(np.ascontiguousarray(cps, dtype=np.uint32).view(np.int32)).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).view(np.int64)).to(dev)
    sc = torch.empty(len(test), dtype=torch.float64, device=dev)
    wsw = torch.empty(max(1, nv.lib.mc_score_words_workspace_bytes(ms_model._dev._h, len(test), max_len)),
                      dtype=torch.uint8, device=dev)
    ms_model._dev.check()
    ms_w = timed(lambda: ms_model._dev.score_words_device(d_cps, d_off, len(test), max_len, 10 ** 20, sc, wsw))
    cw = ms_model._dev.check()
    wbytes = cw["score_probes"] / 4 * 16 + cps.nbytes + off.nbytes + sc.numel() * 8
    out["subword_backoff_scoring"] = {
        "what": f"{len(test)} test passwords (max {max_len} characters) through calc_ml2p's DFS over segmentations on "
                f"the 100k-line tab-split backoff model ({int(ms_model._flat.info[nv.INFO_NENT])} entries): k_score_words",
        "k_score_words": {"ms": ms_w, "passwords_per_s": len(test) / (ms_w * 1e-3), "bytes": wbytes,
                          "hash_probes_per_password": cw["score_probes"] / 4 / len(test),
                          "frac_of_hbm": wbytes / (ms_w * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    del d_cps, d_off, sc, wsw
    torch.cuda.empty_cache()

    # ---- config 5: one round of the secondary-training driver at --size 1e7 (a host loop around the calls above) ----
    import random

    from montecarlo_amd import backwords_secondary_main as bsm
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, "train.txt"), "w") as f:
            f.write("\n".join(synth.zipf_corpus(100000)) + "\n")
        with open(os.path.join(tmp, "test.txt"), "w") as f:
            f.write("\n".join(synth.zipf_corpus(100000, oov_frac=0.01, stream=1)) + "\n")
        random.seed(5)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):       # (the driver's progress lines: stdout carries the JSON line only)
            bsm.wrapper(["-i", os.path.join(tmp, "train.txt"), "-t", os.path.join(tmp, "test.txt"), "-s",
                         os.path.join(tmp, "out"), "--strategy", "samples", "1", "--size", "10000000"])
        torch.cuda.synchronize()
        secs = time.perf_counter() - t0
        cracked = sum(1 for _ in open(os.path.join(tmp, "out", "cracked-iter-0.txt")))
    out["config5_secondary_round"] = {
        "what": "backwords_secondary_main --strategy samples 1 --size 10000000 on 100k training and 100k test lines: one "
                "round (train, 1e7 samples with their strings deduplicated on the GPU, samples file, rank, score, "
                "cracked set) + the final model's pass, files written; wall clock of the whole driver, host work included",
        "seconds": secs, "cracked_in_round": cracked}
    return out


def cpu_baseline(model, test, n_samples: int):
    """The C restatement of the reference's algorithm (oracle/mc_oracle.c), one host thread, on the
    same workload (by default all of it: n_samples samples + their rank build + the test passwords
    in the same proportion scored and ranked; about 10-20 s).  Only the C calls are timed: flattening
    the model and encoding the test set are outside, as they are outside the GPU's timed region."""
    import ctypes as C

    from oracle import c_oracle
    cm = c_oracle.COracleModel(model.nwords, c_oracle.KIND_BACKOFF, start_units=["\x00"])
    t_n = min(len(test), max(1000, n_samples // 10))
    lut = np.full(0x110000, c_oracle.NONE, dtype=np.uint32)
    for s, i in cm.sym_id.items():
        if len(s) == 1:
            lut[ord(s)] = i
    pw = test[:t_n]
    off = np.zeros(t_n + 1, dtype=np.uint64)
    np.cumsum(np.fromiter((len(p) for p in pw), dtype=np.uint64, count=t_n), out=off[1:])
    ids = np.append(lut[np.frombuffer("".join(pw).encode("utf-32-le", "surrogatepass"), dtype="<u4")], np.uint32(0))
    sc = np.empty(t_n, dtype=np.float64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    t0 = time.perf_counter()
    res = cm.sample(n_samples, 20250304)
    t1 = time.perf_counter()
    srt, pos = c_oracle.rank_build(res["ml2p"])
    t2 = time.perf_counter()
    c_oracle.lib().orc_score(cm._h, vp(ids), vp(off), t_n, vp(sc))
    t3 = time.perf_counter()
    order = np.argsort(sc, kind="stable")                   # parse_file's sorted(..., key=ml2p)
    one_core, cnt = c_oracle.rank_query(srt, pos, sc[order])[:2]
    est = np.where(np.asarray(cnt) > 0, one_core, 1.0)      # ml2p_iter2gc's loop as a running maximum
    if float(est.max()) < 2.0 ** 62:
        idx = np.arange(1, t_n + 1, dtype=np.int64)
        ranks = np.maximum.accumulate(np.maximum(np.ceil(est).astype(np.int64) - idx, 0)) + idx
    else:
        ranks, prev = [], 0
        for e in est.tolist():
            prev = int(np.ceil(max(e, prev + 1)))
            ranks.append(prev)
    assert len(ranks) == t_n
    t4 = time.perf_counter()
    every = cpu_all_cores(cm, ids, off, t_n, n_samples, one_core)
    return {"value": n_samples / (t4 - t0), "unit": "sampled passwords/s (same pipeline)",
            "cores": 1, "kind": "port", "all_cores": every,
            "sample": f"{n_samples} samples + rank build + {t_n} test pwds scored, ordered, bisected and ranked, oracle/mc_oracle.c "
                      f"(+ numpy for the order and the running maximum), "
                      f"{t4 - t0:.1f} s of one core",
            "sampling_only_per_s": n_samples / (t1 - t0), "scored_per_s": t_n / (t3 - t2),
            "rank_build_s": t2 - t1, "reference_python_samples_per_s_build_container": 22988}


def cpu_all_cores(cm, ids, off, t_n: int, n_samples: int, one_core):
    """The same C restatement on every host core this job may use, one thread per core (the C calls
    drop the GIL), sharded as the GPUs are: test passwords scored in slices, each thread samples a
    contiguous id range, ranks it against the global N and answers every query, partial positions
    summed; `one_core` = the one-thread positions the sum is compared with."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from oracle import c_oracle
    share = len(os.sched_getaffinity(0))
    p = max(1, min(share, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or 16))
    sc = np.empty(t_n, dtype=np.float64)
    cut = [t_n * w // p for w in range(p + 1)]
    ids_at = ids.ctypes.data

    def score(w):
        o = off[cut[w]:cut[w + 1] + 1].copy()
        base = int(o[0])
        o -= o[0]
        c_oracle.lib().orc_score(cm._h, C.c_void_p(ids_at + 4 * base), o.ctypes.data_as(C.c_void_p),
                                 cut[w + 1] - cut[w], C.c_void_p(sc.ctypes.data + 8 * cut[w]))

    def shard(w):
        lo, hi = n_samples * w // p, n_samples * (w + 1) // p
        res = cm.sample(hi - lo, 20250304, first_id=lo)
        srt, pos = c_oracle.rank_build(res["ml2p"], n_global=n_samples)
        return c_oracle.rank_query(srt, pos, x)[0]

    with ThreadPoolExecutor(p) as pool:
        t0 = time.perf_counter()
        list(pool.map(score, range(p)))
        x = np.sort(sc)
        total = np.zeros(t_n)
        for part in pool.map(shard, range(p)):
            total += part
        t1 = time.perf_counter()
    scale = np.maximum(np.abs(one_core), 1e-300)
    return {"value": n_samples / (t1 - t0), "cores": p, "seconds": t1 - t0,
            "max_rel_diff_vs_one_core": float(np.max(np.abs(total - one_core) / scale)),
            "how": "one thread per core, id-range shards, partial positions summed"}


if __name__ == "__main__":
    main()
