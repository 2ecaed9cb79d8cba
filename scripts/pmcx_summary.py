#!/usr/bin/env python3
"""Per-kernel mean of every counter found in <out>/pmcx_*/ (rocprofv3 --pmc CSV output)."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(out, "pmcx_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            if not name.startswith("k_"):
                continue
            a = acc[name][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n, v = acc[k][c]
        print(f"   {c:28s} {v / n:16.1f}  (mean of {n})")
