#!/usr/bin/env python3
"""kernel_stats.csv of a rocprofv3 run -> one line per kernel (average us, calls)."""
import csv
import sys

for path in sys.argv[1:]:
    print("==", path)
    for r in csv.DictReader(open(path)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if "at::native" in name or "rocclr" in name:
            continue
        print(f"  {name[:44]:44s} calls={r['Calls']:>4s} avg={float(r['AverageNs']) / 1e3:8.1f} us  min={float(r['MinNs']) / 1e3:7.1f}  max={float(r['MaxNs']) / 1e3:7.1f}")
