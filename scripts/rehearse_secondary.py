#!/usr/bin/env python3
"""Rehearsal of BASELINE config 5 over several ranks: `backwords_secondary_main --strategy samples 2`
with the golden run's inputs and streams, launched as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node R --master-addr 127.0.0.1 --master-port P \\
        scripts/rehearse_secondary.py
(on a 1-GPU box the ranks share the GPU and talk over gloo; on an R-GPU node it is RCCL).  Rank 0
checks its result files against the reference's own run (tests/golden/secondary_main.json) -- the sample
ids are global, so a sharded run must write what a single GPU writes."""
import json
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TQDM_DISABLE", "1")

from montecarlo_amd import backwords_secondary_main as main  # noqa: E402
from secondary_check import check_saved  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
with open(os.path.join(GOLDEN, "secondary_main.json")) as f:
    fx = json.load(f)
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
save = os.path.join(tempfile.mkdtemp(prefix="mc-rehearse-"), "save")
random.seed(fx["meta"]["seed"] if rank == 0 else 12345 + rank)      # use_group makes the ranks agree
main.wrapper(["-i", os.path.join(GOLDEN, fx["meta"]["train"]), "-t", os.path.join(GOLDEN, fx["meta"]["test"]),
              "-s", save] + fx["meta"]["args"])
if rank == 0:
    check_saved(save, fx)
    print(f"rehearse_secondary ok: {world} rank(s), files {sorted(os.listdir(save))}", flush=True)
import torch.distributed as dist  # noqa: E402

if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
