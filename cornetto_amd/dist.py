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


_pinned = {}


def _pinned_buf(torch, nbytes, key):
    """cached pinned host staging (rank `dst` receives world x max-bytes per call)"""
    t = _pinned.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        _pinned[key] = t
    return t


def gather_records(arr, global_ctg, device=None, group=None, dst=0, concat=True):
    """Gather a structured record array (field "ctg" = local contig index, rows ordered by it) to rank `dst`.

    global_ctg[i] is the global index of this rank's local contig i.  On `dst` the result has "ctg" rewritten
    to global indices and rows ordered by global contig (stable: the per-contig order each rank produced is
    kept) — the order the reference prints in.  With concat=False the per-rank parts are returned as a list
    of views (no extra copy; they are already in global order whenever every rank owns a contiguous range
    of global indices, as in the one-assembly-per-rank runs).  Other ranks get None."""
    import torch
    import torch.distributed as dist
    arr = np.ascontiguousarray(arr)
    gmap = np.asarray(global_ctg, dtype=np.int64)
    out = arr.copy()
    if len(out):
        out["ctg"] = gmap[arr["ctg"]].astype(arr.dtype["ctg"])
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        res = out[np.argsort(out["ctg"], kind="stable")] if len(out) else out
        return res if concat else [res]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    item = out.dtype.itemsize
    # [count, lowest global ctg, highest global ctg] of every rank
    meta = torch.tensor([len(out), int(out["ctg"][0]) if len(out) else 0, int(out["ctg"][-1]) if len(out) else -1],
                        dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = [[int(x) for x in m.tolist()] for m in metas]
    counts = [m[0] for m in metas]
    mx = max(counts)
    buf = torch.zeros(max(mx, 1) * item, dtype=torch.uint8, device=device)
    if len(out):
        buf[: out.nbytes].copy_(torch.from_numpy(out.view(np.uint8).reshape(-1)), non_blocking=True)
    recv = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dst, group=group)
    if rank != dst:
        return None
    host = _pinned_buf(torch, world * max(mx, 1) * item, (item, "gather"))
    parts = []
    for r in range(world):
        seg = host[r * max(mx, 1) * item: r * max(mx, 1) * item + counts[r] * item]
        seg.copy_(recv[r][: counts[r] * item], non_blocking=True)
        parts.append(seg)
    if buf.is_cuda:
        torch.cuda.current_stream().synchronize()
    parts = [p.numpy().view(out.dtype) for p in parts]
    # already globally ordered when the ranks' contig ranges do not interleave
    spans = [(m[1], m[2]) for m in metas if m[0] > 0]
    ordered = all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
    if not concat and ordered:
        return parts
    allr = np.concatenate(parts) if parts else out
    if not ordered:
        allr = allr[np.argsort(allr["ctg"], kind="stable")]
    return allr if concat else [allr]
