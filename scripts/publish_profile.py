#!/usr/bin/env python3
"""Copy one scripts/gpu_round.sh result (gpurun_out/<tag>) into profiles/ under a round-and-letter name and refresh
profiles/hbm_traffic_latest.json, which ties the HBM traffic bench.py quotes to what was measured: the commit, the
digest of the native sources (bench.py compares it with the tree it runs from), the kernels' average durations of the
kernel-trace run of the same command.  Run in the build container (it asks git for the commit).
Usage: scripts/publish_profile.py <tag> <name, e.g. r02_a>"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from montecarlo_amd.build import sources_digest  # noqa: E402


def main(tag, name):
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    for a, b in (("bench.json", "bench.json"), ("bench_prof.json", "bench_under_profiler.json"),
                 ("kernel_stats.csv", "kernel_stats.csv"), ("pmc_summary.json", "pmc_hbm_traffic.json"),
                 ("other_configs_kernel_stats.csv", "other_configs_kernel_stats.csv"),
                 ("bench_others.json", "bench_other_configs_under_profiler.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, f"{name}_{b}"))
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "montecarlo_amd/csrc", "include"],
                           capture_output=True, text=True).stdout.strip()
    valu = os.path.join(src, "valu_summary.json")           # (scripts/valu_pass.sh)
    if os.path.exists(valu):
        v = json.load(open(valu))
        if v.get("csrc_sha256") != sources_digest():
            sys.exit("the native sources changed since the VALU passes: measure again before publishing")
        v["commit"] = head + (" + uncommitted changes under csrc/ (the digest is of those)" if dirty else "")
        v["source"] = f"profiles/{name}_valu.json"
        for target in (f"{name}_valu.json", "valu_latest.json"):
            json.dump(v, open(os.path.join(dst, target), "w"), indent=1)
    pmc = os.path.join(src, "pmc_summary.json")
    if not os.path.exists(pmc):
        return
    per_step = json.load(open(pmc)).get("_per_step")
    if not per_step:
        sys.exit("no per-step traffic in " + pmc)
    if per_step.get("csrc_sha256") != sources_digest():
        sys.exit("the native sources changed since the measurement: measure again before publishing")
    per_step["commit"] = head + (" + uncommitted changes under csrc/ (the digest below is of those)" if dirty else "")
    per_step["source"] = f"profiles/{name}_pmc_hbm_traffic.json"
    stats = os.path.join(src, "kernel_stats.csv")
    if os.path.exists(stats):
        per_step["kernel_avg_us"] = {
            r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]: round(float(r["AverageNs"]) / 1e3, 2)
            for r in csv.DictReader(open(stats)) if "at::" not in r["Name"] and "rocclr" not in r["Name"]}
        per_step["kernel_stats"] = f"profiles/{name}_kernel_stats.csv"
    json.dump(per_step, open(os.path.join(dst, "hbm_traffic_latest.json"), "w"), indent=1)
    print(json.dumps(per_step, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
