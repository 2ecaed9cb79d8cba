#!/usr/bin/env python3
"""Tuning experiment (GPU box): time k_sample alone on the bench model for a grid of
MC_SAMPLE_CHAINS x MC_SAMPLE_WG_PER_CU x MC_GUIDE_FACTOR x MC_SAMPLE_HOT_KB.  Each configuration runs in a child
process (the library reads its knobs once).  Usage: exp_sample.py [chains,...] [wg,...] [gf,...] [hot_kb,...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
os.environ.setdefault("TQDM_DISABLE", "1")
import torch
from montecarlo_amd import synth
from montecarlo_amd.backwords_simulator import BackWordsMonteCarlo
N = 10_000_000
m = BackWordsMonteCarlo(synth.as_file(synth.zipf_corpus(100000)), splitter="", threshold=10)
dm = m._dev
keys = torch.empty(N, dtype=torch.float64, device="cuda")
for _ in range(3):
    dm.sample(N, 1, out=keys)
torch.cuda.synchronize(); dm.check()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
ev[0].record()
for i in range(10):
    dm.sample(N, 1, first_id=i * N, out=keys); ev[i + 1].record()
torch.cuda.synchronize()
cnt = dm.check()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))
print(json.dumps({"ms_med": ts[5], "ms_min": ts[0], "draws_per_sample": cnt["draws"] / (10 * N), "restarts": cnt.get("restarts"), "indirect_per_draw": cnt["indirect_draws"] / cnt["draws"],
                  "probes_per_draw": cnt["cdf_probes"] / cnt["draws"], "table_mb": m._flat.info[8] / 1e6,
                  "checksum": float(keys.sum())}))
""" % ROOT


def main():
    chains = sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2", "4"]
    wgs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["8"]
    gfs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["4"]
    hots = sys.argv[4].split(",") if len(sys.argv) > 4 else ["20"]
    for gf in gfs:
      for hot in hots:
        for wg in wgs:
            for nc in chains:
                env = dict(os.environ, MC_SAMPLE_CHAINS=nc, MC_SAMPLE_WG_PER_CU=wg, MC_GUIDE_FACTOR=gf,
                           MC_SAMPLE_HOT_KB=hot)
                r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    print(f"chains={nc} wg={wg} gf={gf} hot={hot}: FAILED rc={r.returncode}\n{r.stderr[-2000:]}", flush=True)
                    sys.exit(1)
                print(f"chains={nc} wg={wg} gf={gf} hot={hot}: {r.stdout.strip().splitlines()[-1]}", flush=True)


if __name__ == "__main__":
    main()
