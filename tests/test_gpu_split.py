"""GPU: a contig larger than the fair share cut into pieces with halos (cornetto_amd.dist.SplitPlan) — every "rank"'s pieces through the HIP
kernels as sequences of their own, records cut down to what the piece owns, put together as rank 0 does: equal to the scan of the whole contigs
by the same kernels AND to the CPU oracle.  One GPU plays the ranks one after the other (the exchange itself: tests/test_dist_gloo.py)."""
import numpy as np
import pytest

import oracle_bind as ob
from test_dist_gloo import HIT_DT, IVL_DT, REG_DT, WIN_DT, _cat, _whole, split_case

pytestmark = pytest.mark.gpu


def _gpu_scan(acc, seqs, depth, mq, thr):
    """sequences as one resident assembly + coverage: (hits, wins, ivls, all windows) with ctg = index into seqs"""
    asm = acc.asm_upload(seqs)
    cov = acc.cov_upload(depth, mq)
    try:
        hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
        ivls = acc.sdust(asm, 20, 64)
        sums = acc.cov_prepare(cov, 2500, 50)
        regs = []
        for i in range(len(seqs)):
            r = acc.cov_regs(cov, i)
            a = np.zeros(len(r), dtype=REG_DT)
            a["ctg"] = i
            for k in ("st", "end", "depth", "mq_depth"):
                a[k] = r[k]
            regs.append(a)
        return hits.astype(HIT_DT), wins.astype(WIN_DT), ivls.astype(IVL_DT), _cat(regs, REG_DT), sums
    finally:
        asm.close()
        cov.close()


@pytest.mark.parametrize("world", [2, 3])
def test_pieces_with_halos_equal_the_whole_contigs(world):
    import cornetto_amd
    from cornetto_amd.dist import SplitPlan, make_clean, order_records, stitch_intervals
    lens, seqs, depth, mq, ideal = split_case()
    plan = SplitPlan(lens, world, clean=make_clean(lambda ci, lo, hi: seqs[ci][lo:hi]), min_piece=100000, min_ctg_len=100000)
    assert plan.any_split
    acc = cornetto_amd.Accel(0)
    thr = acc.telowin_threshold(0.4, 99.9)
    try:
        whole = _gpu_scan(acc, seqs, depth, mq, thr)
        parts, tot = [[], [], [], []], np.zeros(3, dtype=np.int64)
        for rank in range(world):
            pcs = plan.pieces[rank]
            if not pcs:
                continue
            h, w, iv, rg, sums = _gpu_scan(acc, [seqs[p[0]][p[3]:p[4]] for p in pcs], [depth[p[0]][p[3]:p[4]] for p in pcs],
                                           [mq[p[0]][p[3]:p[4]] for p in pcs], thr)
            tot += np.array(sums, dtype=np.int64)
            # the halos' sums leave the totals again: a coverage object over the halo ranges alone
            hr = plan.halo_ranges(rank)
            if hr:
                hc = acc.cov_upload([depth[c][a:b] for c, a, b in hr], [mq[c][a:b] for c, a, b in hr])
                tot -= np.array(acc.cov_prepare(hc, 2500, 50), dtype=np.int64)
                hc.close()
            gl = np.array(plan.global_ctg(rank), dtype=np.int32)
            for k, a in enumerate((plan.own_points(rank, h, "start"), plan.own_points(rank, w, "start"), plan.own_intervals(rank, iv),
                                   plan.own_points(rank, rg, "st"))):
                a["ctg"] = gl[a["ctg"]]
                parts[k].append(a)
    finally:
        acc.close()
    got = [order_records(_cat(parts[0], HIT_DT), ("strand", "start")), order_records(_cat(parts[1], WIN_DT), ("start",)),
           stitch_intervals(_cat(parts[2], IVL_DT)), order_records(_cat(parts[3], REG_DT), ("st",))]
    exp = _whole(lens, seqs, depth, mq, ob.telowin_threshold(0.4, 99.9))
    if world == 2:                                    # (the cut of the two-rank plan lies inside the planted (AC)n array: an interval crosses it)
        c = plan.cuts[0][0]
        assert any(r[0] == 0 and r[1] < c < r[2] for r in exp[2].tolist())
    for name, g, w_, e in zip(("telofind", "telowin", "sdust", "coverage windows"), got, whole[:4], exp):
        assert np.array_equal(w_, e), name + ": whole contigs on the GPU against the oracle"
        assert np.array_equal(g, e), name + ": pieces against the oracle"
    assert tuple(int(x) for x in tot) == tuple(int(x) for x in whole[4])
