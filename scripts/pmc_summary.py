#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (one run each) per kernel.

On ROCm 7.2 the derived counters FETCH_SIZE / WRITE_SIZE are in kilobytes (1024 B).  MI355X_MICROARCH.md (HBM section):
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports half of the bytes of wide
coalesced reads -> doubled here (`fetch_bytes_corrected`); WRITE_SIZE is exact.  Narrow random
gathers are uncalibrated by the guide, so both raw and corrected figures are kept.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                name = name.split("(")[0].strip()
                per[name][0] += 1
                per[name][1] += float(row["Counter_Value"])
    return per


def main(out):
    res = {}
    fetch = load(os.path.join(out, "pmc_FETCH_SIZE"))
    write = load(os.path.join(out, "pmc_WRITE_SIZE"))
    for k in sorted(set(fetch) | set(write)):
        # the same dispatch is reported once per XCD/instance on some versions: keep calls as counted
        fc, fv = fetch.get(k, [0, 0.0])
        wc, wv = write.get(k, [0, 0.0])
        res[k] = {"calls": max(fc, wc),
                  "fetch_kb_per_call_raw": fv / fc if fc else None,
                  "fetch_bytes_per_call_corrected": 2 * 1024 * fv / fc if fc else None,
                  "write_bytes_per_call": 1024 * wv / wc if wc else None}
    # HBM bytes per bench step: k_sample's launch, and all launches of one mc_rank_build together
    steps = max((v["calls"] for k, v in res.items() if k.startswith("k_sample")), default=0)
    if steps:
        def total(pred):
            t = 0.0
            for k, v in res.items():
                if pred(k):
                    t += v["calls"] * ((v["fetch_bytes_per_call_corrected"] or 0.0) + (v["write_bytes_per_call"] or 0.0))
            return t / steps
        rank_kernels = ("k_hist_all", "k_plan", "k_copy_if_unsorted", "k_onesweep", "k_weight_sums", "k_scan_partials",
                        "k_positions")
        latest = {"k_sample": total(lambda k: k.startswith("k_sample")),
                  "mc_rank_build": total(lambda k: k.split("<")[0] in rank_kernels),
                  "k_score": total(lambda k: k.startswith("k_score")), "k_query": total(lambda k: k == "k_query"),
                  "note": "HBM bytes per bench step (FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE), rocprofv3 --pmc"}
        res["_per_step"] = latest
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
