"""Multi-GPU plumbing for the contig-sharded panel path: one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).

The scans are independent per contig, so the data path needs no collective.  What is exchanged:
  * the contig -> rank assignment (computed identically on every rank from the contig lengths: no message);
  * ONE all-reduce of 3 x int64 {sum depth, sum mq, positions}: the assembly-wide mean the (no)boringbits
    thresholds are derived from (src/boringbits_main.c:293-294 -> :518-519);
  * the gather of result records (BED/TSV rows as fixed-size structs) to rank 0, which restores the
    reference's print order (input contig order).
"""
import numpy as np


def lpt_partition(lengths, world):
    """longest-processing-time-first bin packing of contigs over `world` ranks.
    Returns a list (per rank) of ascending global contig indices; deterministic, so every rank computes
    the same table without communication."""
    lengths = [int(x) for x in lengths]
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    parts = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += lengths[i]
    return [sorted(p) for p in parts]


def allreduce_sums(sums, device=None, group=None):
    """all-reduce (sum) of the three exact integer totals of cornetto_cov_prepare()"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(x) for x in sums], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
    return tuple(int(x) for x in t.tolist())


def gather_records(arr, global_ctg, device=None, group=None, dst=0):
    """Gather a structured record array (field "ctg" = local contig index) to rank `dst`.

    global_ctg[i] is the global index of this rank's local contig i.  On `dst` the result is one array with
    "ctg" rewritten to global indices and rows ordered by global contig (stable: the per-contig order each
    rank produced is kept) — the order the reference prints in.  Other ranks get None."""
    import torch
    import torch.distributed as dist
    arr = np.ascontiguousarray(arr)
    out = arr.copy()
    if len(out):
        out["ctg"] = np.asarray(global_ctg, dtype=np.int64)[arr["ctg"]].astype(arr.dtype["ctg"])
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return out[np.argsort(out["ctg"], kind="stable")] if len(out) else out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    item = out.dtype.itemsize
    n = torch.tensor([len(out)], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    buf = torch.zeros(max(mx, 1) * item, dtype=torch.uint8, device=device)
    if len(out):
        buf[: out.nbytes] = torch.from_numpy(out.view(np.uint8).reshape(-1).copy()).to(buf.device)
    recv = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [recv[r][: counts[r] * item].cpu().numpy().view(out.dtype) for r in range(world)]
    allr = np.concatenate(parts) if parts else out
    return allr[np.argsort(allr["ctg"], kind="stable")]
