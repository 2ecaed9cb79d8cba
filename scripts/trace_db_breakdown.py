#!/usr/bin/env python3
"""Per-kernel time of single rank builds out of a rocprofv3 --kernel-trace database (the default .db output):
  trace_db_breakdown.py <results.db> <build index> [...]   (builds are counted by their first k_sniff / k_sniff_count)"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
def short(n):
    m = re.search(r"(k_\w+|__amd\w+|at::native::\w+)", n); return m.group(1) if m else n[:40]
seq = [(short(n), (e - s) / 1e3) for n, s, e in rows]
first = "k_sniff_count" if any(n == "k_sniff_count" for n, _ in seq) else "k_sniff"
idx = [i for i, (n, _) in enumerate(seq) if n == first]
for which in [int(x) for x in sys.argv[2:]]:
    a = idx[which]; b = idx[which + 1] if which + 1 < len(idx) else len(seq)
    agg = {}
    for n, d in seq[a:b]:
        agg.setdefault(n, [0, 0.0]); agg[n][0] += 1; agg[n][1] += d
    print("build", which, "total us", sum(v[1] for v in agg.values()))
    for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"   {n:28s} x{c:3d} {d:10.1f} us")
