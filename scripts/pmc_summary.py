#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (one run each) per kernel and per stage of a bench step.

On ROCm 7.2 the derived counters FETCH_SIZE / WRITE_SIZE are in kilobytes (1024 B).  MI355X_MICROARCH.md (HBM section):
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports half of the bytes of wide
coalesced reads -> doubled here (`fetch_bytes_corrected`); WRITE_SIZE is exact.  Narrow random
gathers are uncalibrated by the guide, so both raw and corrected figures are kept.

The sort's kernels run three times a step (mc_rank_build, and twice inside mc_order_keys): the per-stage totals
follow the dispatches in order (k_sample opens a step, k_positions or k_rl_final closes the rank build, k_order_prep .. k_order_out
bracket the order; since r02-m, which dropped k_order_prep, the order starts with the first kernel after k_score).  `_per_step` also carries the digest of the native sources the numbers were measured on.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0].strip()


def rows(dirname):
    out = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                out.append((int(row["Dispatch_Id"]), short(row["Kernel_Name"]), float(row["Counter_Value"])))
    out.sort()
    return out


def per_kernel(rs):
    per = defaultdict(lambda: [0, 0.0])
    for _, name, v in rs:
        per[name][0] += 1
        per[name][1] += v
    return per


def per_stage(rs):
    """kilobytes per stage, summed over the steps seen, and the number of steps"""
    tot = defaultdict(float)
    stage, steps = None, 0
    # the sampling launches after the timed region (bench.py's roofline.alone) are no steps: cut where two of our
    # kernels in a row are k_sample
    ours = [i for i, (_, name, _) in enumerate(rs) if not name.startswith("at::") and "rocclr" not in name]
    for a, b in zip(ours, ours[1:]):
        if rs[a][1].split("<")[0] == "k_sample" and rs[b][1].split("<")[0] == "k_sample":
            rs = rs[:a]
            break
    for _, name, v in rs:
        base = name.split("<")[0]
        if base == "k_sample":
            stage, steps = "k_sample", steps + 1
        elif base.startswith("k_score"):
            stage = "k_score"
        elif stage == "k_score" and not base.startswith("k_score"):
            stage = "mc_order_keys"          # (the first kernel after the scoring: the order's key sort)
        elif base == "k_query":
            stage = "k_query"
        elif base.startswith("k_gn_") or base == "k_query_chunk_max":   # (mc_rank_runs_guess_numbers: the bisection inside)
            stage = "mc_guess_numbers"
        elif stage == "k_sample":
            stage = "mc_rank_build"          # (the first kernel after the sampler: the sort's memset or k_sniff)
        if stage is None or base.startswith("at::") or "rocclr_copy" in base:
            continue
        tot[stage] += v
        if base in ("k_positions", "k_rl_final"):
            stage = "after_rank_build"
        elif base.startswith("k_order_out"):
            stage = "after_order"
    return tot, steps


def main(out):
    res = {}
    fr, wr = rows(os.path.join(out, "pmc_FETCH_SIZE")), rows(os.path.join(out, "pmc_WRITE_SIZE"))
    fetch, write = per_kernel(fr), per_kernel(wr)
    for k in sorted(set(fetch) | set(write)):
        fc, fv = fetch.get(k, [0, 0.0])
        wc, wv = write.get(k, [0, 0.0])
        res[k] = {"calls": max(fc, wc),
                  "fetch_kb_per_call_raw": fv / fc if fc else None,
                  "fetch_bytes_per_call_corrected": 2 * 1024 * fv / fc if fc else None,
                  "write_bytes_per_call": 1024 * wv / wc if wc else None}
    fs, fsteps = per_stage(fr)
    ws, wsteps = per_stage(wr)
    if fsteps and wsteps:
        latest = {k: 2 * 1024 * fs.get(k, 0.0) / fsteps + 1024 * ws.get(k, 0.0) / wsteps
                  for k in sorted(set(fs) | set(ws)) if not k.startswith("after_")}
        latest["write_bytes"] = {k: 1024 * ws.get(k, 0.0) / wsteps for k in sorted(ws) if not k.startswith("after_")}
        latest["note"] = ("HBM bytes per bench step and stage (FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE), "
                          "rocprofv3 --pmc, one pass per counter")
        try:
            from montecarlo_amd.build import sources_digest
            latest["csrc_sha256"] = sources_digest()
        except Exception as e:  # noqa: BLE001
            latest["csrc_sha256"] = None
            latest["digest_error"] = str(e)
        res["_per_step"] = latest
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
