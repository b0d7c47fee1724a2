"""world_size-2 gloo test of the N>1 path's host logic: the PRODUCT's list plan (skh_list_plan_owners: items dealt to ranks
by size -- the plan skh_scan_list follows; whole files here, SK_NO_SPLIT=1, because the per-rank counts come from the
oracle, which scans files) + the agreement on the plan the ranks reach before scanning (dist.plans_agree) + the count
all-reduce identity (sum over ranks of shard counts == unsharded counts).  A CPU test of the distributed plumbing, not
of the kernels."""
import os
import sys

import numpy as np
import torch.multiprocessing as tmp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _worker(rank, world, port, files, strain, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    import _oracle
    import strainer2_amd as sk
    from strainer2_amd.dist import allreduce_count_array, plans_agree
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SK_NO_SPLIT"] = "1"
    os.environ["SK_THREADS"] = str(1 + 4 * rank)                       # (the plan must not depend on it)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    list_path = os.path.join(out_dir, "list.txt")
    assert plans_agree(list_path, world)
    owners = sk.KmerContext.list_plan_owners(list_path, world)
    assert len(owners) == len(files) and set(owners.tolist()) == set(range(world))
    t = _oracle.OracleTable()
    assert t.build_file(strain) == 0
    mine = [f for i, f in enumerate(files) if owners[i] == rank]
    for f in mine:
        t.scan_file(f, 2)
    _keys, counts = t.rows()
    col = np.ascontiguousarray(counts[:, 2])
    col[0] += np.uint32(0xFFFFFFFF) if rank == 0 else np.uint32(2)      # wrap-around must survive the sum
    total = allreduce_count_array(col)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), total)
    dist.destroy_process_group()


def test_sharded_allreduce_equals_unsharded(golden, tmp_path):
    sys.path.insert(0, HERE)
    import _oracle
    d = os.path.join(golden, "cases", "mixed")
    files = [os.path.join(d, f) for f in ["m1.fasta", "m2.fq.gz", "m3_crlf.fa", "g1.fa", "g2.fa.gz"]]
    strain = os.path.join(d, "strain.fna.gz")
    port = 29500 + os.getpid() % 2000
    with open(tmp_path / "list.txt", "w") as f:
        f.write("".join(p + "\n" for p in files))
    tmp.spawn(_worker, args=(2, port, files, strain, str(tmp_path)), nprocs=2, join=True)
    t = _oracle.OracleTable()
    t.build_file(strain)
    for f in files:
        t.scan_file(f, 2)
    want = t.rows()[1][:, 2].copy()
    want[0] += np.uint32(1)                                             # 0xFFFFFFFF + 2 wraps to +1
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)
    assert np.array_equal(r0, want)
    assert want.sum() > 0
