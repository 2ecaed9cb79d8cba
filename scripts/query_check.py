import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from montecarlo_amd.engine import DeviceRank
g = torch.Generator().manual_seed(3)
dev = "cuda"
bad = 0
for n in (1, 5, 4096, 4097, 100003, 10_000_019):
    keys = (torch.rand(n, generator=g, dtype=torch.float64) * 60).to(dev)
    if n > 100: keys[n // 3: n // 2] = 17.25          # a long run
    rk = DeviceRank(keys)
    srt, pos = rk.keys_sorted, rk.positions
    for name, x in {
        "sorted dense": torch.sort(torch.rand(1_000_001, generator=g, dtype=torch.float64) * 62 - 1)[0],
        "unsorted": torch.rand(300_007, generator=g, dtype=torch.float64) * 62 - 1,
        "ties and edges": torch.tensor([17.25, 0.0, -0.0, -5.0, 1e9, float("inf"), -float("inf"), 17.25] * 40 + [float(srt[0]), float(srt[-1])], dtype=torch.float64),
        "with a nan": torch.tensor([3.0, float("nan"), 50.0] * 30, dtype=torch.float64),
        "exact keys": srt[:: max(1, n // 5000)].cpu().clone(),
        "one": torch.tensor([20.0], dtype=torch.float64),
    }.items():
        x = x.to(dev).contiguous()
        qp, qc = rk.query(x)
        want_c = torch.searchsorted(srt, x, right=True)
        want_c = torch.where(torch.isnan(x), torch.full_like(want_c, n), want_c)
        want_p = torch.where(want_c > 0, pos[(want_c - 1).clamp(min=0)], torch.zeros_like(x))
        ok = torch.equal(qc, want_c) and torch.equal(qp.view(torch.int64), want_p.view(torch.int64))
        bad += not ok
        print(n, name, "ok" if ok else "BAD", flush=True)
print("ALL OK" if not bad else "%d BAD" % bad)
sys.exit(1 if bad else 0)
