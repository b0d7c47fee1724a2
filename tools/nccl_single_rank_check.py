import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
import strainer2_amd as sk
from strainer2_amd import synth
from strainer2_amd.dist import allreduce_counts
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
contigs = synth.make_strain()
ks = sk.Keyset.from_stream(synth.strain_stream(contigs))
ctx = sk.KmerContext(0); ctx.load_keyset(ks, 4)
reads, nb = synth.make_reads(contigs, 200000, 150, seed=1)
ctx.scan_stream(reads, 2); ctx.sync()
before = ctx.counts(2).copy()
allreduce_counts(ctx, 2)
allreduce_counts(ctx)
after = ctx.counts(2)
assert np.array_equal(before, after) and before.sum() > 0
print("nccl single-rank all-reduce on the library's counter block: ok", int(before.sum()))
dist.destroy_process_group()
