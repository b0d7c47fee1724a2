"""Multi-GPU plumbing: one process per GPU, table replicated, inputs sharded, and ONE sum
all-reduce of the per-k-mer count block at the end (RCCL over xGMI via torch.distributed;
gloo on CPU for tests).  New relative to the reference, which is single-process.
"""
import numpy as np


class _DevBlock:
    """Expose a raw HIP device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def plans_agree(list_path, world, skip=None) -> bool:
    """Every rank of the default process group computed the same work plan for `list_path` (skh_list_plan_hash: items
    dealt by size, big plain files cut into byte ranges -- a function of the list, the file sizes and the world size).
    Call before scan_list(rank=.., world=..) when the counters are reduced through torch.distributed; the programs'
    own RCCL path (skh_scan_list with sk_comm_init) makes the same comparison inside the library."""
    import torch
    import torch.distributed as dist
    from .native import KmerContext
    h = KmerContext.list_plan_hash(list_path, world, skip)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([h & 0x7FFFFFFF, (h >> 31) & 0x7FFFFFFF, h >> 62], dtype=torch.int64, device=dev)
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool((lo == hi).all().item())


def scan_list_sharded(ctx, list_path, col, rank, world, skip=None) -> int:
    """This rank's share of a -A/-B/-C list into column `col` for callers that reduce through torch.distributed (the
    programs' own RCCL path does all of this inside skh_scan_list).  Every rank issues the same collectives whatever happens
    to it locally: (1) the plans must agree (plans_agree; ValueError on every rank otherwise); (2) if the plan cuts a file
    into byte ranges, every rank keeps a copy of its column; (3) after the scan one MIN all-reduce of "my scan was fine / a cut
    did not hold / it failed otherwise"; (4) a cut that did not hold anywhere (inputs the reference accepts,
    src/genome_compare.c:203): every rank puts its column back and scans its share of the whole-file plan; any other failure:
    OSError on every rank.  Returns the bases this rank scanned."""
    import torch
    import torch.distributed as dist
    from .native import SK_E_SPLIT, SKError, KmerContext
    if world > 1 and not plans_agree(list_path, world, skip):
        raise ValueError(f"the ranks computed different work plans for {list_path} (skh_list_plan_hash)")
    shared = world > 1 and bool((KmerContext.list_plan_owners(list_path, world, skip) == 0xFFFFFFFE).any())
    keep = ctx.counts(col) if shared else None
    status, bases, why = 2, 0, ""
    try:
        bases = ctx.scan_list(list_path, col, skip=skip, rank=rank, world=world)
    except SKError as e:
        status, why = (1 if e.code == SK_E_SPLIT and shared else 0), str(e)
    if world > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([status], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        worst = int(t.item())
    else:
        worst = status
    if worst == 0:
        raise OSError(f"a rank's scan of {list_path} failed" + (f": {why}" if why and status == 0 else ""))
    if worst == 1:                                        # a cut did not hold somewhere: everybody goes round again, uncut
        ctx.set_counts(col, keep)
        ok = 1
        try:
            bases = ctx.scan_list(list_path, col, skip=skip, rank=rank, world=world, uncut=True)
        except SKError as e:
            ok, why = 0, str(e)
        if world > 1:
            t = torch.tensor([ok], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        if not ok:
            raise OSError(f"a rank's uncut scan of {list_path} failed" + (f": {why}" if why else ""))
    return bases


def allreduce_counts(ctx, col=None):
    """In-place sum over the default process group of ctx's counter block -- all columns, or only
    column `col` (the per-k-mer count vector of the list that was scanned).  The block is in the
    device's locality order, identical on every rank that loaded the same key set.
    u32 wrap-around == i32 two's-complement sum, so it is reduced as int32."""
    import torch
    import torch.distributed as dist
    ctx.sync()
    if col is None:
        ptr, n = ctx.counts_device_ptr(), ctx.nrows * ctx.ncols
    else:
        ptr, n = ctx.counts_device_ptr() + 4 * col * ctx.nrows, ctx.nrows
    t = torch.as_tensor(_DevBlock(ptr, n), device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()


def allreduce_count_array(counts: np.ndarray):
    """Host-array variant (gloo): used by the CPU tests of the sharding identity."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(counts.view(np.int32).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy().view(np.uint32)
