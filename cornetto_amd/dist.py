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


# ---- contigs larger than the fair share: pieces with halos -------------------------------------------------------------
#
# Every scan of the path is a LOCAL function of the contig around a position, as long as the stretch around a cut is plain sequence:
#   sdust      the result is the union of the perfect intervals (each at most W bases long, a function of the words inside it); what the
#              reference carries from base to base — the word window, its v suffix, the list P (src/sdust/sdust.c:66-128,130-160) — is a
#              function of the last 62 words, the gate of :149 of the 62 before those, so 2 (W - 2) words of plain letters in front of a
#              position fix its state; another byte (N) leaves the window stale (:152-156) and reaches W bases further.  A cut with HALO
#              plain letters on either side therefore splits a contig into two scans whose interval sets, clipped to the cut and merged
#              again where they touch (the rule of :94-98), are the one scan's.
#   telofind   a run (src/find_telomere.c:44-58) is a chain of adjacent matches; with no match of the motif or its reverse complement
#              within HALO of the cut no run crosses it and the greedy search passes the cut in its start state.
#   telowin    windows of 1000 at multiples of 200 (src/telomere_windows.c:28-43): the piece that holds a window's start owns it and needs the
#              marks of 1000 bases behind it -> cuts at multiples of 200, halo >= 1000.
#   coverage   windows [j inc, j inc + w) (src/boringbits_main.c:322-378): owned by the piece that holds the start; cuts at multiples of inc,
#              halo >= w behind; the three sums behind the mean (:283-294) count every position once: the halos' sums are subtracted.
# Cuts are multiples of lcm(64, 200, inc) (the resident layout wants 64-byte aligned starts), the halo is the next multiple of that above
# max(1600, w, 4 W); a cut position is only taken where `clean(ctg, lo, hi)` says the bases [lo, hi) are plain ACGT (either case) without a
# motif match — every rank asks the same question of the same bases, so every rank computes the same table without a message.

def _lcm(a, b):
    from math import gcd
    return a // gcd(a, b) * b


class SplitPlan:
    """pieces[rank] = [(ctg, start, end, lo, hi)]: the rank scans bases [lo, hi) of contig ctg as one sequence and owns what starts in
    [start, end); a contig that is not cut is one piece with lo = start = 0 and hi = end = its length"""

    def __init__(self, lengths, world, clean=None, window=2500, inc=50, W=64, search=0.10, min_piece=1 << 20, min_ctg_len=1000000, tol=0.05):
        # (a piece is never shorter than -m: print_fun_bits' length test, src/boringbits_main.c:428, must see a piece as it sees its contig.  The
        # deprecated `boringbits` selection tests window positions against the contig's ends, :473: not for pieces)
        min_piece = max(int(min_piece), int(min_ctg_len) + 1)
        self.lengths = [int(x) for x in lengths]
        self.world = int(world)
        self.gran = _lcm(_lcm(64, 200), max(1, int(inc)))
        need = max(1600, int(window), 4 * int(W))
        self.halo = (need + self.gran - 1) // self.gran * self.gran
        total = sum(self.lengths)
        fair = total / float(max(1, world))
        self.cuts = {ci: [] for ci in range(len(self.lengths))}
        whole = [(ci, 0, n, 0, n) for ci, n in enumerate(self.lengths)]
        lpt = lpt_partition(self.lengths, self.world)
        lpt_max = max(sum(self.lengths[i] for i in p) for p in lpt) if self.lengths else 0
        if self.world <= 1 or clean is None or lpt_max <= (1.0 + tol) * fair:
            # whole contigs pack well enough (the HG002 assembly: 0.7 % over the fair share on 4 ranks, 3.8 % on 8; 22 % on 16): nothing is cut
            self.pieces = [[whole[i] for i in p] for p in lpt]
        else:
            # the contigs in input order on one line, rank r takes [r, r + 1) x fair of it: a border inside a contig becomes a cut on the nearest
            # clean position; one closer than min_piece to a contig's end moves to that end
            starts = np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)
            reach = int(search * fair) // self.gran
            borders = []                                        # (ctg, position in it): rank r + 1 begins there
            for r in range(1, self.world):
                x = int(round(r * fair))
                ci = int(np.searchsorted(starts, x, "right")) - 1
                ci = min(max(ci, 0), len(self.lengths) - 1)
                n, off = self.lengths[ci], x - int(starts[ci])
                prev = borders[-1] if borders else (0, 0)
                if prev[0] > ci:
                    # the border before this one found no clean cut and moved to the END of this contig: this one cannot lie in front of it (found by
                    # tests/test_dist_gloo.py::test_split_plan_properties_on_random_assemblies: the piece behind the earlier border owned the contig's
                    # tail, and so did the piece behind this cut)
                    borders.append(prev)
                    continue
                lo_ok = (prev[1] if prev[0] == ci else 0) + max(min_piece, 2 * self.halo)
                c = None
                if off >= lo_ok and n - off >= max(min_piece, 2 * self.halo):
                    ideal = int(round(off / float(self.gran))) * self.gran
                    for d in range(0, reach + 1):
                        for cand in ((ideal,) if d == 0 else (ideal + d * self.gran, ideal - d * self.gran)):
                            if cand < lo_ok or cand > n - max(min_piece, 2 * self.halo):
                                continue
                            if clean(ci, cand - self.halo, cand + self.halo):
                                c = cand
                                break
                        if c is not None:
                            break
                if c is not None:
                    self.cuts[ci].append(c)
                    borders.append((ci, c))
                else:                                           # no cut: the border moves to the nearer end of the contig (not in front of the last border)
                    b = (ci + 1, 0) if (off * 2 >= n or (prev[0] == ci and prev[1] > 0)) else (ci, 0)
                    borders.append(max(b, prev))
            borders = [(0, 0)] + borders + [(len(self.lengths), 0)]
            self.pieces = []
            for r in range(self.world):
                (c0, p0), (c1, p1) = borders[r], borders[r + 1]
                mine = []
                for ci in range(c0, min(c1 + 1, len(self.lengths))):
                    n = self.lengths[ci]
                    s_ = p0 if ci == c0 else 0
                    e_ = p1 if ci == c1 else n
                    if e_ > s_ or (n == 0 and ci < c1):
                        mine.append((ci, s_, e_, s_ - self.halo if s_ > 0 else 0, e_ + self.halo if e_ < n else n))
                self.pieces.append(mine)
        self.loads = [sum(p[2] - p[1] for p in pp) for pp in self.pieces]
        self.any_split = any(len(c) > 0 for c in self.cuts.values())

    # ---- what a rank does with the records of its pieces (local "ctg" = index into pieces[rank]) ----
    def own_points(self, rank, arr, key):
        """records owned by where they START (telomere runs, telomere windows, coverage windows): keep the rows whose `key` field, moved to
        contig coordinates, lies in [start, end) of their piece; "ctg" stays the local piece index, coordinates become the contig's"""
        P = np.array([(p[1], p[2], p[3]) for p in self.pieces[rank]], dtype=np.int64).reshape(-1, 3)
        if len(arr) == 0:
            return arr.copy()
        li = arr["ctg"].astype(np.int64)
        shift = P[li, 2]
        pos = arr[key].astype(np.int64) + shift
        keep = (pos >= P[li, 0]) & (pos < P[li, 1])
        out = arr[keep].copy()
        sh = shift[keep]
        for f in out.dtype.names:
            if f in ("start", "end", "finish", "st"):
                out[f] = (out[f].astype(np.int64) + sh).astype(out.dtype[f])
        return out

    def own_intervals(self, rank, ivls):
        """sdust intervals clipped to [start, end) of their piece, in contig coordinates (a piece that is a whole contig keeps the reference's
        intervals beyond the contig's end: nothing is clipped there)"""
        if len(ivls) == 0:
            return ivls.copy()
        P = np.array([(p[1], p[2], p[3], p[4], self.lengths[p[0]]) for p in self.pieces[rank]], dtype=np.int64).reshape(-1, 5)
        li = ivls["ctg"].astype(np.int64)
        st = ivls["start"].astype(np.int64) + P[li, 2]
        fi = ivls["finish"].astype(np.int64) + P[li, 2]
        lo = np.where(P[li, 0] > 0, P[li, 0], np.int64(-1) << 40)                 # a cut on this side clips, a contig end does not
        hi = np.where(P[li, 1] < P[li, 4], P[li, 1], np.int64(1) << 40)
        st2, fi2 = np.maximum(st, lo), np.minimum(fi, hi)
        keep = fi2 > st2
        out = ivls[keep].copy()
        out["start"] = st2[keep].astype(out.dtype["start"])
        out["finish"] = fi2[keep].astype(out.dtype["finish"])
        return out

    def global_ctg(self, rank):
        """a sortable global key per local piece: contig-major, then the piece's start (what gather_records orders by)"""
        return [p[0] for p in self.pieces[rank]]

    def halo_ranges(self, rank):
        """[(ctg, lo, hi)] the positions a rank scans without owning them: their sums leave the three totals again"""
        out = []
        for ci, s, e, lo, hi in self.pieces[rank]:
            if lo < s:
                out.append((ci, lo, s))
            if hi > e:
                out.append((ci, e, hi))
        return out


def stitch_intervals(ivls):
    """rank 0, after the gather (rows ordered by contig, then by start: stable over the ranks' pieces): two intervals of one contig that touch or
    overlap — the two halves of an interval a cut went through — become one, by the reference's own rule (src/sdust/sdust.c:94-98)"""
    if len(ivls) < 2:
        return ivls
    order = np.lexsort((ivls["start"], ivls["ctg"]))
    a = ivls[order]
    join = (a["ctg"][1:] == a["ctg"][:-1]) & (a["start"][1:] <= a["finish"][:-1])
    if not join.any():
        return a
    head = np.concatenate([[True], ~join])
    grp = np.cumsum(head) - 1
    out = a[head].copy()
    fin = np.zeros(len(out), dtype=np.int64)
    np.maximum.at(fin, grp, a["finish"].astype(np.int64))
    out["finish"] = fin.astype(out.dtype["finish"])
    return out


def order_records(arr, keys):
    """rank 0, after the gather of records of pieces: the reference's print order again — by contig, then `keys` (telomere runs: strand, start —
    find() prints the forward search, then the reverse one, src/find_telomere.c:44-74; telomere and coverage windows: start)"""
    if len(arr) < 2:
        return arr
    return arr[np.lexsort(tuple(arr[k] for k in reversed(keys)) + (arr["ctg"],))]


def make_clean(seq_of, motif=b"TTAGGG"):
    """clean(ctg, lo, hi) for SplitPlan over `seq_of(ctg, lo, hi) -> uint8 array` (host or device memory, a file window ...): plain ACGT in
    either case and no match of the motif or of its reverse complement (src/find_telomere.c:24-42)"""
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    rc = motif.translate(comp)[::-1]
    ok = np.zeros(256, dtype=bool)
    ok[list(b"ACGTacgt")] = True

    def clean(ci, lo, hi):
        s = np.asarray(seq_of(ci, lo, hi), dtype=np.uint8)
        if len(s) != hi - lo or not ok[s].all():
            return False
        b = s.tobytes().upper()
        return motif not in b and rc not in b
    return clean
