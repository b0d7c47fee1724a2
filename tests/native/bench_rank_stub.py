"""A stand-in for a rank of bench.py (TEST CODE, tests/test_bench_launcher.py): joins the gloo group the launcher set up,
all-reduces one number, prints noise on stdout from every rank and -- rank 0 -- one result line.  --steps 99: rank 1 fails."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--warmup", type=int, default=0)
ap.add_argument("--backend", default="gloo")
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == args.gpus and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([rank + 1], dtype=torch.int64)
dist.all_reduce(t)
print(f"noise on stdout from rank {rank} {{not json", flush=True)
if args.steps == 99 and rank == 1:
    sys.exit(3)
dist.barrier()
if rank == 0:
    print(json.dumps({"metric": "stub", "value": int(t.item()), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                      "port": os.environ["MASTER_PORT"]}), flush=True)
dist.destroy_process_group()
