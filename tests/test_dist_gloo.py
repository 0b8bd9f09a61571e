"""CPU, world_size 2 over gloo: the multi-GPU plumbing (contig partition, the one all-reduce, the ordered
gather of records to rank 0).  The per-contig compute is replaced by the CPU oracle here — this test is
about the exchange, the GPU kernels are covered by test_gpu_parity.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from cornetto_amd.dist import allreduce_sums, gather_records, lpt_partition  # noqa: E402

IVL_DT = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("finish", "<i4")])


def _contigs():
    rng = np.random.default_rng(42)
    lens = [5000, 120, 0, 900, 15000, 64, 3000, 7000, 1, 2500, 11000]
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seqs = []
    for n in lens:
        s = alpha[rng.integers(0, 4, size=n)].copy()
        if n > 500:
            s[100:400] = np.frombuffer((b"AC" * 150), dtype=np.uint8)
            s[450:460] = ord("N")
        seqs.append(s)
    return seqs


def _local_sdust(seqs, idxs):
    import oracle_bind as ob
    rows = []
    for li, gi in enumerate(idxs):
        for r in ob.sdust(seqs[gi], 20, 64):
            rows.append((li, int(r) >> 32, int(r) & 0xFFFFFFFF))
    return np.array(rows, dtype=IVL_DT) if rows else np.zeros(0, dtype=IVL_DT)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = _contigs()
    parts = lpt_partition([len(s) for s in seqs], world)
    mine = parts[rank]
    recs = _local_sdust(seqs, mine)
    allr = gather_records(recs, mine)
    sums = allreduce_sums((sum(len(seqs[i]) for i in mine), rank + 1, len(mine)))
    if rank == 0:
        q.put((allr.tolist(), sums))
    dist.barrier()
    dist.destroy_process_group()


def test_lpt_partition_is_a_balanced_partition():
    lens = [242_000_000, 200_000_000, 150_000_000, 90_000_000, 61_000_000, 5_000_000, 100_000, 90_000, 0, 7]
    for world in (1, 2, 3, 4, 8):
        parts = lpt_partition(lens, world)
        assert sorted(i for p in parts for i in p) == list(range(len(lens)))
        loads = [sum(lens[i] for i in p) for p in parts]
        assert max(loads) <= sum(lens) / world + max(lens)
        assert all(p == sorted(p) for p in parts)
    assert lpt_partition(lens, 2) == lpt_partition(lens, 2)      # deterministic: no message needed


@pytest.mark.timeout(300)
def test_two_ranks_gather_restores_reference_order():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, sums = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = _contigs()
    exp = _local_sdust(seqs, list(range(len(seqs))))             # single-process order = reference print order
    assert got == exp.tolist()
    assert sums == (sum(len(x) for x in seqs), 3, len(seqs))


def test_single_process_gather_is_identity_order():
    recs = np.array([(1, 5, 9), (0, 1, 2), (1, 20, 30)], dtype=IVL_DT)
    out = gather_records(recs, [7, 3])
    assert out.tolist() == [(3, 5, 9), (3, 20, 30), (7, 1, 2)]
    assert allreduce_sums((1, 2, 3)) == (1, 2, 3)


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cornetto_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_reads_shares_cut_the_stream_at_read_boundaries(world):
    """config 5 (src/seq.c:116-129 per read): the shares of the ranks, in rank order, are the stream; no read is split,
    none is in two shares, and the shares are balanced by bases to within one read"""
    bench = _bench_module()
    rng = np.random.default_rng(7)
    pieces = [np.clip(rng.lognormal(9.2, 0.9, size=n), 200, 200000).astype(np.int64) for n in (700, 650)]
    n_pieces = 5
    shares = bench.reads_shares(pieces, n_pieces, world)
    assert len(shares) == world
    stream = [(j % 2, i) for j in range(n_pieces) for i in range(len(pieces[j % 2]))]
    got = [(s, i) for sh in shares for (s, lo, hi) in sh for i in range(lo, hi)]
    assert got == stream
    B = sum(int(pieces[j % 2].sum()) for j in range(n_pieces))
    per = [sum(int(pieces[s][lo:hi].sum()) for s, lo, hi in sh) for sh in shares]
    assert sum(per) == B
    assert max(abs(p - B / world) for p in per) <= 2 * 200000
    assert all(hi > lo for sh in shares for _, lo, hi in sh)


def test_reads_shares_with_more_ranks_than_reads():
    bench = _bench_module()
    shares = bench.reads_shares([np.array([1000, 10], dtype=np.int64)], 1, 4)
    assert [(s, i) for sh in shares for (s, lo, hi) in sh for i in range(lo, hi)] == [(0, 0), (0, 1)]


def test_plain_bench_with_several_gpus_starts_its_own_ranks_or_refuses():
    """`python bench.py --gpus 2` (the shape of the driver's command) never needs a launcher around it: without two visible
    GPUs it prints no line and exits 3 (here: no GPU at all); the launch itself is covered on the GPU box
    (tests/test_gpu_bench_ranks.py::test_plain_command_starts_its_own_ranks)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the launch is tested by tests/test_gpu_bench_ranks.py")
    assert p.returncode == 3, p.stderr.decode(errors="replace")[-2000:]
    assert p.stdout == b"" and b"refusing" in p.stderr


# ---- contigs larger than the fair share: pieces with halos (cornetto_amd.dist.SplitPlan) -----------------------------------------------

HIT_DT = np.dtype([("ctg", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("end", "<i4")])
WIN_DT = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("end", "<i4"), ("car", "<i4")])
REG_DT = np.dtype([("ctg", "<i4"), ("st", "<i4"), ("end", "<i4"), ("depth", "<i4"), ("mq_depth", "<i4")])


def split_case(scale=1):
    """three contigs, the first 70 % of the bases; around the middle of the first (where a two-rank plan wants its cut): an N run on the ideal
    position, a telomere array in front of it (no clean position on that side for 12 kb), an (AC)n array behind it — the first clean position lies
    INSIDE that array, so sdust intervals cross the cut and must be put together again; N runs, poly-A and lower case elsewhere"""
    rng = np.random.default_rng(2024)
    lens = [4_200_000 * scale, 1_000_000 * scale, 800_000 * scale]
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs, depth, mq = [], [], []
    for n in lens:
        s = acgt[rng.integers(0, 4, size=n)].copy()
        for p in range(20000, n - 20000, 150000):
            kind = (p // 150000) % 4
            if kind == 0:
                s[p:p + 240] = ord("A")
            elif kind == 1:
                s[p:p + 300] = ord("N")
                s[p + 310:p + 400] = np.frombuffer(b"AC" * 45, dtype=np.uint8)
            elif kind == 2:
                s[p:p + 600] = np.frombuffer(b"TTAGGG" * 100, dtype=np.uint8)
            else:
                s[p:p + 500] |= 0x20
        s[:6000] = np.frombuffer(b"CCCTAA" * 1000, dtype=np.uint8)
        s[n - 3000:] = np.frombuffer(b"TTAGGG" * 500, dtype=np.uint8)
        seqs.append(s)
        d = rng.poisson(30, size=(n + 999) // 1000).repeat(1000)[:n].astype(np.int64) + rng.integers(-2, 3, size=n)
        d = np.clip(d, 0, 65535)
        q = d.copy()
        for p in range(40000, n - 70000, 90000):
            if (p // 90000) % 2:
                d[p:p + 7000] //= 5
                q[p:p + 7000] = d[p:p + 7000]
            else:
                q[p + 20000:p + 31000] //= 4
        depth.append(d.astype(np.uint16))
        mq.append(q.astype(np.uint16))
    ideal = int(round(sum(lens) / 2.0 / 1600)) * 1600            # where a two-rank plan wants its border: half of the bases, inside contig 0
    s = seqs[0]
    s[ideal - 50:ideal + 50] = ord("N")
    s[ideal - 12000:ideal - 4002] = np.frombuffer(b"TTAGGG" * 1333, dtype=np.uint8)
    s[ideal + 3000:ideal + 12000] = np.frombuffer(b"AC" * 4500, dtype=np.uint8)
    return lens, seqs, depth, mq, ideal


def oracle_scan(seq, d, q, thr):
    """one sequence through the CPU oracle: (hits, wins, sdust intervals, all coverage windows) with ctg = 0"""
    import oracle_bind as ob
    oh = ob.telofind(seq, b"TTAGGG")
    hits = np.array([(0, int(h["strand"]), int(h["start"]), int(h["end"])) for h in oh], dtype=HIT_DT)
    wins = np.array([(0, int(w["start"]), int(w["end"]), int(w["car"])) for w in ob.telowin(oh, len(seq), thr)], dtype=WIN_DT)
    ivls = np.array([(0, int(x) >> 32, int(x) & 0xFFFFFFFF) for x in ob.sdust(seq, 20, 64)], dtype=IVL_DT)
    r = ob.get_regs(d, q, 2500, 50)
    regs = np.zeros(len(r), dtype=REG_DT)
    for k in ("st", "end", "depth", "mq_depth"):
        regs[k] = r[k]
    return hits, wins, ivls, regs


def _cat(parts, dt):
    parts = [p for p in parts if len(p)]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=dt)


def scan_pieces(plan, rank, seqs, depth, mq, thr, scan=oracle_scan):
    """what a rank does: its pieces as sequences of their own, the records moved to contig coordinates and cut down to what the piece owns;
    the three sums without the halos"""
    H, Wn, I, R = [], [], [], []
    sums = np.zeros(3, dtype=np.int64)
    for li, (ci, s, e, lo, hi) in enumerate(plan.pieces[rank]):
        h, w, iv, rg = scan(seqs[ci][lo:hi], depth[ci][lo:hi], mq[ci][lo:hi], thr)
        for a in (h, w, iv, rg):
            a["ctg"] = li
        H.append(h), Wn.append(w), I.append(iv), R.append(rg)
        sums += (int(depth[ci][lo:hi].astype(np.int64).sum()), int(mq[ci][lo:hi].astype(np.int64).sum()), hi - lo)
    for ci, a, b in plan.halo_ranges(rank):
        sums -= (int(depth[ci][a:b].astype(np.int64).sum()), int(mq[ci][a:b].astype(np.int64).sum()), b - a)
    return (plan.own_points(rank, _cat(H, HIT_DT), "start"), plan.own_points(rank, _cat(Wn, WIN_DT), "start"),
            plan.own_intervals(rank, _cat(I, IVL_DT)), plan.own_points(rank, _cat(R, REG_DT), "st"), sums)


def _whole(lens, seqs, depth, mq, thr):
    out = [[], [], [], []]
    for ci in range(len(lens)):
        for k, a in enumerate(oracle_scan(seqs[ci], depth[ci], mq[ci], thr)):
            a["ctg"] = ci
            out[k].append(a)
    return [_cat(out[0], HIT_DT), _cat(out[1], WIN_DT), _cat(out[2], IVL_DT), _cat(out[3], REG_DT)]


def _split_worker(rank, world, port, q):
    from cornetto_amd.dist import SplitPlan, make_clean, order_records, stitch_intervals
    import oracle_bind as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens, seqs, depth, mq, ideal = split_case()
    plan = SplitPlan(lens, world, clean=make_clean(lambda ci, lo, hi: seqs[ci][lo:hi]), min_piece=100000, min_ctg_len=100000)
    thr = ob.telowin_threshold(0.4, 99.9)
    hits, wins, ivls, regs, sums = scan_pieces(plan, rank, seqs, depth, mq, thr)
    gl = plan.global_ctg(rank)
    g = [gather_records(a, gl) for a in (hits, wins, ivls, regs)]
    tot = allreduce_sums(sums)
    if rank == 0:
        res = [order_records(g[0], ("strand", "start")), order_records(g[1], ("start",)), stitch_intervals(g[2]), order_records(g[3], ("st",))]
        q.put(([r.tolist() for r in res], tot, plan.cuts, [[tuple(int(x) for x in p) for p in pp] for pp in plan.pieces], ideal))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_a_contig_larger_than_the_fair_share_is_cut_and_two_ranks_equal_one():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, tot, cuts, pieces, ideal = q.get(timeout=500)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import oracle_bind as ob
    lens, seqs, depth, mq, _ = split_case()
    # the plan: contig 0 (70 % of the bases) in two pieces, the cut on the first clean position — inside the (AC)n array behind the N run
    assert cuts[0] == [ideal + 4800] and cuts[1] == [] and cuts[2] == []
    loads = [sum(p[2] - p[1] for p in pp) for pp in pieces]
    assert max(loads) < 0.52 * sum(lens) and sorted(p[0] for pp in pieces for p in pp) == [0, 0, 1, 2]
    exp = _whole(lens, seqs, depth, mq, ob.telowin_threshold(0.4, 99.9))
    assert len(exp[0]) > 50 and len(exp[1]) > 20 and len(exp[2]) > 100 and len(exp[3]) > 100000
    c = cuts[0][0]
    assert any(r[0] == 0 and r[1] < c < r[2] for r in exp[2].tolist())        # an sdust interval of the one scan crosses the cut
    for name, g, e in zip(("telofind", "telowin", "sdust", "coverage windows"), got, exp):
        assert g == e.tolist(), name
    assert tot == (sum(int(d.astype(np.int64).sum()) for d in depth), sum(int(x.astype(np.int64).sum()) for x in mq), sum(lens))


def test_split_plan_leaves_an_assembly_without_large_contigs_alone_and_is_deterministic():
    from cornetto_amd.dist import SplitPlan
    from cornetto_amd.synth import contig_lengths
    lens = contig_lengths(0)                                  # the HG002 assembly of the bench (100 contigs, the largest 242 Mb)
    for world in (1, 2, 4, 8, 16):
        plan = SplitPlan(lens, world, clean=lambda ci, lo, hi: True)
        flat = sorted(p for pp in plan.pieces for p in pp)
        if world <= 8:
            assert not plan.any_split
            assert [sorted(p[0] for p in pp) for pp in plan.pieces] == lpt_partition(lens, world)
        else:                                                 # whole contigs: 22 % over the fair share
            assert plan.any_split and max(plan.loads) < 1.001 * sum(lens) / world
        # a partition of every contig into consecutive owned ranges
        for ci, n in enumerate(lens):
            own = [(p[1], p[2]) for p in flat if p[0] == ci]
            assert own[0][0] == 0 and own[-1][1] == n and all(own[k][1] == own[k + 1][0] for k in range(len(own) - 1))
    few = [250_000_000, 240_000_000, 100_000_000]
    plan = SplitPlan(few, 8, clean=lambda ci, lo, hi: (lo // 1600) % 3 != 0)         # (two of three candidate positions are refused)
    assert plan.any_split and max(plan.loads) < 1.15 * sum(few) / 8
    assert plan.pieces == SplitPlan(few, 8, clean=lambda ci, lo, hi: (lo // 1600) % 3 != 0).pieces
    for pp in plan.pieces:
        for ci, s, e, lo, hi in pp:
            assert s % plan.gran == 0 and (e % plan.gran == 0 or e == few[ci]) and lo == max(0, s - plan.halo) and hi == min(few[ci], e + plan.halo)


def test_split_plan_properties_on_random_assemblies():
    """randomised: whatever the contig lengths and the number of ranks, the pieces of a SplitPlan own every base of every contig exactly once, in contig
    order over the ranks; a piece's scan range reaches one halo beyond a cut and stops at a contig's end; cuts lie on the granule (64-base tiles, 200-base
    telomere windows, the window step), no piece is shorter than -m, only clean positions are cut, and the plan is a function of its arguments"""
    import numpy as np
    from cornetto_amd.dist import SplitPlan
    rng = np.random.default_rng(20261004)
    n_cut = 0
    for it in range(400):
        nctg = int(rng.integers(1, 30))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            lens = rng.integers(1, 5_000_000, size=nctg)
        elif kind == 1:
            lens = np.concatenate([[int(rng.integers(50_000_000, 250_000_000))], rng.integers(1000, 20_000_000, size=nctg)])
        elif kind == 2:
            lens = np.array([int(rng.integers(100_000_000, 300_000_000))] * int(rng.integers(1, 4)) + [int(x) for x in rng.integers(0, 3_000_000, size=nctg)])
        else:
            lens = (10 ** rng.uniform(3, 8.3, size=nctg)).astype(np.int64)
        lens = [int(x) for x in lens]
        world = int(rng.choice([1, 2, 3, 4, 7, 8, 16]))
        inc = int(rng.choice([50, 1, 49, 130, 1000]))
        window = int(rng.choice([2500, 300, 5000]))
        dirty = {(int(c), int(p)) for c, p in zip(rng.integers(0, len(lens), size=40), rng.integers(0, 300_000_000, size=40))}

        def clean(ci, lo, hi, dirty=dirty):
            return not any(c == ci and lo <= p < hi for c, p in dirty)
        plan = SplitPlan(lens, world, clean=clean, window=window, inc=inc, W=64, min_ctg_len=1000000)
        again = SplitPlan(lens, world, clean=clean, window=window, inc=inc, W=64, min_ctg_len=1000000)
        assert plan.pieces == again.pieces
        assert len(plan.pieces) == world
        owned = {ci: [] for ci in range(len(lens))}
        last = (-1, -1)
        for r in range(world):
            for ci, s, e, lo, hi in plan.pieces[r]:
                n = lens[ci]
                assert 0 <= lo <= s <= e <= hi <= n, (lens, world, plan.pieces[r])
                if plan.any_split:                                                  # (cut: the contigs in input order over the ranks; not cut: dealt by length)
                    assert (ci, s) > last or (e == s and n == 0)
                last = (ci, s)
                owned[ci].append((s, e))
                assert lo == (s - plan.halo if s > 0 else 0) and hi == (e + plan.halo if e < n else n)
                if s > 0:
                    assert s % plan.gran == 0 and clean(ci, s - plan.halo, s + plan.halo)
                if (s, e) != (0, n):
                    assert e - s > 1000000 and e - s >= 2 * plan.halo                # a piece of a cut contig is longer than -m
                    n_cut += 1
        for ci, n in enumerate(lens):
            spans = sorted(owned[ci])
            assert spans and spans[0][0] == 0 and spans[-1][1] == n, (ci, n, spans)
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])), (ci, spans)   # no gap, no overlap
        assert sum(plan.loads) == sum(lens)
    assert n_cut > 50


def test_pieces_give_back_the_whole_contigs_records_on_random_plans():
    """randomised: the records of a whole contig — intervals (sdust: some cross a cut, touch it, reach beyond the contig's end) and records owned by where
    they start (telomere runs, windows) — seen through the pieces of a SplitPlan (every piece "scans" its [lo, hi) and reports what it finds there, in
    its own coordinates), cut down by own_intervals / own_points, gathered in rank order and stitched, are the whole contig's records again"""
    import numpy as np
    from cornetto_amd.dist import SplitPlan, stitch_intervals, order_records
    IVL = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("finish", "<i4")])
    HIT = np.dtype([("ctg", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("end", "<i4")])
    rng = np.random.default_rng(777)
    n_cross = 0
    for it in range(200):
        lens = [int(x) for x in rng.integers(2_000_000, 40_000_000, size=int(rng.integers(1, 6)))]
        lens.insert(int(rng.integers(0, len(lens) + 1)), int(rng.integers(80_000_000, 200_000_000)))
        world = int(rng.choice([2, 3, 4, 8]))
        plan = SplitPlan(lens, world, clean=lambda ci, lo, hi: True)
        cuts = {ci: sorted(c) for ci, c in plan.cuts.items()}
        # the whole contigs' records: disjoint, non-touching intervals (as sdust's union gives them), some placed across and at the cuts
        ivls, hits = [], []
        for ci, n in enumerate(lens):
            pts = sorted(set(int(x) for x in rng.integers(0, n - 200, size=60)) | {c - int(rng.integers(1, 60)) for c in cuts[ci]} | {c for c in cuts[ci][:1]})
            last = -10
            for p in pts:
                if p <= last + 1 or p < 0:
                    continue
                f = min(n + 30, p + int(rng.integers(7, 150)))
                if cuts[ci] and rng.random() < 0.1:
                    f = max(f, cuts[ci][0] + (0 if rng.random() < 0.5 else 40))       # ends exactly at a cut, or crosses it
                ivls.append((ci, p, f))
                last = f
            ivls.append((ci, max(last + 5, n - 40), n + 25))                            # reaches beyond the contig's end (src/sdust/sdust.c:88-102)
            for p in sorted(set(int(x) for x in rng.integers(0, n - 10, size=80)) | set(cuts[ci]) | {c - 1 for c in cuts[ci]}):
                hits.append((ci, int(rng.integers(0, 2)), p, p + 6))
        whole_i = np.array(sorted(ivls), dtype=IVL)
        # (intervals made above may overlap where a forced end passed the next start: the union, as the reference gives it)
        whole_i = stitch_intervals(whole_i)
        whole_h = order_records(np.array(hits, dtype=HIT), ["strand", "start"])
        got_i, got_h = [], []
        for r in range(world):
            loc_i, loc_h = [], []
            for li, (ci, s, e, lo, hi) in enumerate(plan.pieces[r]):
                n = lens[ci]
                for c, a, b in whole_i[whole_i["ctg"] == ci].tolist():
                    # what a scan of [lo, hi) reports: the part of the interval inside it (beyond the contig's end only for a piece that ends there)
                    a2, b2 = max(a, lo), (b if hi == n else min(b, hi))
                    if b2 > a2:
                        loc_i.append((li, a2 - lo, b2 - lo))
                for c, st, a, b in whole_h[whole_h["ctg"] == ci].tolist():
                    if lo <= a and b <= hi:
                        loc_h.append((li, st, a - lo, b - lo))
            oi = plan.own_intervals(r, np.array(loc_i, dtype=IVL))
            oh = plan.own_points(r, np.array(loc_h, dtype=HIT), "start")
            g = np.array(plan.global_ctg(r), dtype=np.int32)
            if len(oi):
                oi["ctg"] = g[oi["ctg"]]
            if len(oh):
                oh["ctg"] = g[oh["ctg"]]
            got_i.append(oi)
            got_h.append(oh)
        gi = stitch_intervals(np.concatenate(got_i))
        gh = order_records(np.concatenate(got_h), ["strand", "start"])
        assert gi.tolist() == whole_i.tolist(), (lens, world, cuts)
        assert gh.tolist() == whole_h.tolist(), (lens, world, cuts)
        n_cross += sum(1 for c, a, b in whole_i.tolist() if any(a < x < b for x in cuts[c]))
    assert n_cross > 20
