"""panel interval stage (scripts/create-cornetto.sh:44-66).  PARITY UNPINNED against bedtools
(not available here): the product is checked against the oracle's independent restatement and a hand-worked case."""
import numpy as np
import pytest

import cornetto_amd
import oracle_bind as ob


def _o2p(a):
    out = np.zeros(len(a), cornetto_amd.IVL_DT)
    out["ctg"], out["start"], out["finish"] = a["ctg"], a["start"], a["end"]
    return out


def _p2o(a):
    out = np.zeros(len(a), ob.SPAN_DT)
    out["ctg"], out["start"], out["end"] = a["ctg"], a["start"], a["finish"]
    return out


def test_panel_boring_hand_worked():
    """three contigs, defaults of the script.  c0 (2 Mb): fun 500000-540000 and a lowQ row 1000000-1009000 grow by 40 kb,
    the edges are 0-200000 and 1800000-2000000; merge -d 200000 joins 0-200000 with nothing (460000-200000 = 260000),
    460000-580000 stays, 960000-1049000 stays, the last edge stays; what is left: four gaps.  c1 (700 kb) is shorter
    than 800 kb: nothing.  c2 (900 kb): a lowQ row that starts at 30000 (not > 40000: unchanged), 7999 long: dropped."""
    lens = [2_000_000, 700_000, 900_000]
    fun = np.array([(0, 500_000, 540_000)], cornetto_amd.IVL_DT)
    lowq = np.array([(0, 1_000_000, 1_009_000), (2, 30_000, 37_999), (1, 100_000, 300_000)], cornetto_amd.IVL_DT)
    got = cornetto_amd.panel_boring(lens, fun, lowq)
    exp = [(0, 200_000, 460_000), (0, 580_000, 960_000), (0, 1_049_000, 1_800_000), (2, 200_000, 700_000)]
    assert [tuple(int(x) for x in r) for r in got] == exp
    assert np.array_equal(_p2o(got), ob.panel_boring(lens, _p2o(fun), _p2o(lowq)))


@pytest.mark.parametrize("seed", range(12))
def test_panel_boring_random_vs_oracle(seed):
    rng = np.random.default_rng(seed)
    n_ctg = int(rng.integers(1, 30))
    lens = rng.integers(1, 3_000_000, size=n_ctg).astype(np.int32)
    lens[rng.integers(0, n_ctg)] = 800_000
    def rows(k, maxlen):
        out = []
        for _ in range(k):
            c = int(rng.integers(0, n_ctg))
            a = int(rng.integers(0, lens[c]))
            out.append((c, a, min(int(lens[c]), a + int(rng.integers(1, maxlen)))))
        return np.array(out, cornetto_amd.IVL_DT).reshape(-1)
    fun, lowq = rows(int(rng.integers(0, 40)), 200_000), rows(int(rng.integers(0, 40)), 30_000)
    kw = {} if seed % 2 == 0 else dict(min_lowq_len=int(rng.integers(1, 20000)), extend=int(rng.integers(0, 90000)), edge_len=int(rng.integers(1, 400000)),
                                       merge_dist=int(rng.integers(0, 300000)), min_ctg_len=int(rng.integers(0, 1500000)))
    got = cornetto_amd.panel_boring(lens, fun, lowq, **kw)
    exp = ob.panel_boring(lens, _p2o(fun), _p2o(lowq), **kw)
    assert np.array_equal(_p2o(got), exp)
    # the result is a set of disjoint, ordered pieces inside contigs that are long enough
    assert all(0 <= r["start"] < r["finish"] <= lens[r["ctg"]] for r in got)


@pytest.mark.parametrize("seed", range(6))
def test_panel_boring_asymmetric_extend_vs_oracle(seed):
    """the three constants of scripts/recreate-cornetto.sh:38 — `if ($2 > 50000) {$2 - 40000, $3 + 50000}` — are separate options"""
    rng = np.random.default_rng(100 + seed)
    n_ctg = 12
    lens = rng.integers(200_000, 4_000_000, size=n_ctg).astype(np.int32)
    lowq = np.array([(int(c), int(a), int(a + l)) for c, a, l in zip(rng.integers(0, n_ctg, 60), rng.integers(0, 190_000, 60), rng.integers(100, 30_000, 60))],
                    cornetto_amd.IVL_DT)
    fun = np.zeros(0, cornetto_amd.IVL_DT)
    kw = dict(extend=int(rng.integers(0, 60000)), extend_right=int(rng.integers(0, 90000)), extend_gate=int(rng.integers(0, 120000)))
    got = cornetto_amd.panel_boring(lens, fun, lowq, **kw)
    exp = ob.panel_boring(lens, _p2o(fun), _p2o(lowq), **kw)
    assert np.array_equal(_p2o(got), exp)


def test_panel_recreate_script_constants_hand_worked():
    """scripts/recreate-cornetto.sh:34-49, no coverage stage.  One 3 Mb contig: a lowQ row 60000-67500 (7 500 long: kept, starts
    beyond 50 000: becomes 20000-117500), one 50000-58000 (starts AT 50 000, not beyond: unchanged), one 1500000-1507499 (7 499:
    dropped); edges 0-200000 and 2800000-3000000; merge -d 200000 folds the two rows into the left edge; left over: 200000-2800000.
    A 999 999-base contig is shorter than 1 Mb: nothing."""
    lens = [3_000_000, 999_999]
    lowq = np.array([(0, 60_000, 67_500), (0, 50_000, 58_000), (0, 1_500_000, 1_507_499), (1, 400_000, 500_000)], cornetto_amd.IVL_DT)
    got = cornetto_amd.panel_boring(lens, np.zeros(0, cornetto_amd.IVL_DT), lowq, recreate=True)
    assert [tuple(int(x) for x in r) for r in got] == [(0, 200_000, 2_800_000)]
    exp = ob.panel_boring(lens, np.zeros(0, ob.SPAN_DT), _p2o(lowq), 7500, 40000, 200000, 200000, 1000000, 50000, 50000)
    assert np.array_equal(_p2o(got), exp)
    # with the row moved one base further it is extended: 50001 - 40000 = 10001 .. 58001 + 50000
    lowq[1] = (0, 50_001, 58_001)
    lowq2 = np.array([(0, 900_000, 910_000)], cornetto_amd.IVL_DT)
    got = cornetto_amd.panel_boring(lens, np.zeros(0, cornetto_amd.IVL_DT), np.concatenate([lowq, lowq2]), recreate=True)
    # 860000-960000 is more than 200 kb from the left block (ends 200000) and from the right edge: it stays a block of its own
    assert [tuple(int(x) for x in r) for r in got] == [(0, 200_000, 860_000), (0, 960_000, 2_800_000)]


def _hg002_fixture(golden_dir):
    """the reference's own bigenough fixture (test/bigenough/hg002-cornetto-E_3): chroms.bed = the assembly BED of a real HG002
    hifiasm assembly, in.boringbits.bed = what steps 1-9 of scripts/create-cornetto.sh (awk, sort, bedtools merge / subtract) made
    of it — the one piece of real bedtools output the reference holds for this stage"""
    import os
    names, lens = [], []
    for l in open(os.path.join(golden_dir, "bigenough", "chroms.bed")):
        n, _a, b = l.split()
        names.append(n)
        lens.append(int(b))
    idx = {n: i for i, n in enumerate(names)}
    rows = [(idx[l.split()[0]], int(l.split()[1]), int(l.split()[2])) for l in open(os.path.join(golden_dir, "bigenough", "in.boringbits.bed"))]
    return names, np.array(lens, np.int32), rows


def _panel_invariants(lens, rows):
    """what `merge -d 200000`, the 200 kb edges and the 800 kb cut of create-cornetto.sh:55-66 imply for ANY input, read off the
    reference-held bedtools output first: rows ordered by contig then start, disjoint; every row starts at or after 200 000 and
    ends at or before length - 200 000; a row is longer than 200 000 (two fun blocks closer than that were merged); no row on a
    contig shorter than 800 000"""
    prev = None
    for c, a, b in rows:
        assert 0 <= a < b <= lens[c]
        assert a >= 200_000 and b <= lens[c] - 200_000, (c, a, b)
        assert b - a > 200_000, (c, a, b)
        assert lens[c] >= 800_000
        if prev is not None:
            assert (c, a) > (prev[0], prev[2]) or c > prev[0]
            if c == prev[0]:
                assert a > prev[2]
        prev = (c, a, b)


def test_reference_held_bedtools_output_invariants_hold_for_the_product(golden_dir):
    """(1) the invariants are facts of the reference's fixture; (2) cornetto_panel_boring keeps them for random fun / lowQ sets over the
    same assembly; (3) the oracle agrees row for row.  The stage stays UNPINNED (no bedtools run), this is the structural pin."""
    names, lens, rows = _hg002_fixture(golden_dir)
    _panel_invariants(lens, rows)
    assert min(b - a for _c, a, b in rows) == 209_966 and sum(1 for c, a, b in rows if a == 200_000) == 35
    for seed in range(8):
        rng = np.random.default_rng(700 + seed)
        def draw(k, lo, hi):
            out = []
            for _ in range(k):
                c = int(rng.integers(0, len(lens)))
                a = int(rng.integers(0, lens[c]))
                out.append((c, a, min(int(lens[c]), a + int(rng.integers(lo, hi)))))
            return np.array(out, cornetto_amd.IVL_DT).reshape(-1)
        fun, lowq = draw(int(rng.integers(50, 600)), 30_000, 400_000), draw(int(rng.integers(0, 300)), 1, 40_000)
        got = cornetto_amd.panel_boring(lens, fun, lowq)
        _panel_invariants(lens, [(int(r["ctg"]), int(r["start"]), int(r["finish"])) for r in got])
        assert np.array_equal(_p2o(got), ob.panel_boring(lens, _p2o(fun), _p2o(lowq)))


def test_panel_reproduces_the_reference_held_bedtools_output(golden_dir):
    """Solved backwards: the fun blocks are the complement of in.boringbits.bed inside every contig; an inner block (e, s) is what
    one fun row [e + 40000, s - 40000) becomes after step 5's +-40 kb; the block at a contig's start / end is the 200 kb edge
    merged with a row that reaches it within 200 kb.  Fed with those rows, steps 4-9 must give the reference's file exactly."""
    names, lens, rows = _hg002_fixture(golden_dir)
    by_ctg = {}
    for c, a, b in rows:
        by_ctg.setdefault(c, []).append((a, b))
    fun = []
    for c, L in enumerate(lens.tolist()):
        if L < 800_000:
            continue
        v = by_ctg.get(c, [])
        if not v:                                         # (one contig of the fixture: everything is fun)
            fun.append((c, 240_000, L - 240_000))
            continue
        for (_a, e), (s, _b) in zip(v, v[1:]):            # inner blocks
            assert s - e >= 80_000 + 1
            fun.append((c, e + 40_000, s - 40_000))
        S, E = v[0][0], v[-1][1]
        if S > 200_000:                                   # left edge grown to S
            fun.append((c, 240_000 if S - 40_000 > 240_000 else S - 41_000, S - 40_000))
        if E < L - 200_000:                               # right edge grown down to E
            fun.append((c, E + 40_000, L - 240_000 if L - 240_000 > E + 40_000 else E + 41_000))
    fun = np.array(sorted(fun), cornetto_amd.IVL_DT)
    got = cornetto_amd.panel_boring(lens, fun, np.zeros(0, cornetto_amd.IVL_DT))
    assert [(int(r["ctg"]), int(r["start"]), int(r["finish"])) for r in got] == rows
    assert np.array_equal(_p2o(got), ob.panel_boring(lens, _p2o(fun), np.zeros(0, ob.SPAN_DT)))


def _bed(golden_dir, name):
    import os
    rows = [l.split("\t") for l in open(os.path.join(golden_dir, "bedtools", name)).read().splitlines() if l]
    return [(r[0], int(r[1]), int(r[2])) for r in rows]


def test_bedtools_manual_merge_examples(golden_dir):
    """tests/golden/bedtools/: the worked examples of the bedtools manual for `merge` (restated from the manual, not bedtools
    output: see the README there) through the oracle; the device merge runs the same cases in the GPU test below"""
    for fin, fexp, d in (("merge_default.in.bed", "merge_default.exp.bed", 0), ("merge_d1000.in.bed", "merge_d1000.exp.bed", 1000)):
        rows = _bed(golden_dir, fin)
        spans = np.array([(0, a, b) for _n, a, b in rows], ob.SPAN_DT)
        got = ob.ivl_merge(spans, d)
        assert [(int(r["start"]), int(r["end"])) for r in got] == [(a, b) for _n, a, b in _bed(golden_dir, fexp)], fin
    # derived from the manual's wording ("maximum distance between features allowed for features to be merged"):
    spans = np.array([(0, 100, 200), (0, 200 + 7, 300), (0, 300 + 8, 400)], ob.SPAN_DT)     # gap == d merges, gap == d + 1 does not
    assert [(int(r["start"]), int(r["end"])) for r in ob.ivl_merge(spans, 7)] == [(100, 300), (308, 400)]


def test_bedtools_manual_subtract_example(golden_dir):
    """`bedtools subtract -a A -b B` (manual, default behaviour): chr1 10-20 is covered by B 0-30 and disappears, chr1 100-200
    keeps 100-180.  The panel stage only ever subtracts from whole contigs (create-cornetto.sh:62,66): the example is run with
    each A row as a contig of its own whose coordinates start at the row's start."""
    a, b, exp = (_bed(golden_dir, "subtract_default.%s.bed" % k) for k in ("a", "b", "exp"))
    out = []
    for name, a0, a1 in a:
        # B clipped to the A row and shifted so that the row is the contig [0, a1 - a0)
        inside = [(0, max(b0, a0) - a0, min(b1, a1) - a0) for n, b0, b1 in b if n == name and min(b1, a1) > max(b0, a0)]
        got = ob.panel_boring([a1 - a0], np.array(inside, ob.SPAN_DT).reshape(-1), np.zeros(0, ob.SPAN_DT), 1, 0, 1 << 30, 0, 0, 0, 1 << 30)
        prod = cornetto_amd.panel_boring([a1 - a0], _o2p(np.array(inside, ob.SPAN_DT).reshape(-1)), np.zeros(0, cornetto_amd.IVL_DT),
                                         min_lowq_len=1, extend=0, extend_right=0, extend_gate=1 << 30, edge_len=1 << 30, merge_dist=0, min_ctg_len=0)
        assert np.array_equal(_p2o(prod), got)
        out += [(name, int(r["start"]) + a0, int(r["end"]) + a0) for r in got]
    assert out == exp
    # derived: several B per A, a B inside A, a book-ended B
    lens = [1000]
    bs = np.array([(0, 0, 30), (0, 180, 300), (0, 300, 310), (0, 500, 600)], ob.SPAN_DT)
    got = ob.panel_boring(lens, bs, np.zeros(0, ob.SPAN_DT), 1, 0, 1 << 30, 0, 0, 0, 1 << 30)
    assert [(int(r["start"]), int(r["end"])) for r in got] == [(30, 180), (310, 500), (600, 1000)]


@pytest.fixture(scope="module")
def acc():
    a = cornetto_amd.Accel(0)
    yield a
    a.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,dist", [(1, 1, 0), (2, 50, 0), (3, 5000, 10), (4, 200_000, 1000), (5, 1_000_000, 0), (6, 70_000, 200_000)])
def test_ivl_merge_vs_oracle(acc, seed, n, dist):
    rng = np.random.default_rng(seed)
    ctg = np.sort(rng.integers(0, max(1, n // 1000 + 3), size=n)).astype(np.int32)
    start = rng.integers(0, 5_000_000, size=n).astype(np.int32)
    order = np.lexsort((start, ctg))
    iv = np.zeros(n, cornetto_amd.IVL_DT)
    iv["ctg"], iv["start"] = ctg[order], start[order]
    iv["finish"] = iv["start"] + rng.integers(0, 3000, size=n).astype(np.int32)
    if n > 10:
        iv["finish"][n // 2] = iv["start"][n // 2] + 4_000_000            # one interval that swallows many
    got = acc.ivl_merge(iv, dist)
    assert np.array_equal(_p2o(got), ob.ivl_merge(_p2o(iv), dist))


@pytest.mark.gpu
def test_device_merge_on_the_bedtools_manual_examples(acc, golden_dir):
    for fin, fexp, d in (("merge_default.in.bed", "merge_default.exp.bed", 0), ("merge_d1000.in.bed", "merge_d1000.exp.bed", 1000)):
        rows = _bed(golden_dir, fin)
        got = acc.ivl_merge(np.array([(0, a, b) for _n, a, b in rows], cornetto_amd.IVL_DT), d)
        assert [(int(r["start"]), int(r["finish"])) for r in got] == [(a, b) for _n, a, b in _bed(golden_dir, fexp)], fin
    got = acc.ivl_merge(np.array([(0, 100, 200), (0, 207, 300), (0, 308, 400)], cornetto_amd.IVL_DT), 7)
    assert [(int(r["start"]), int(r["finish"])) for r in got] == [(100, 300), (308, 400)]


@pytest.mark.gpu
def test_ivl_merge_rejects_unordered_input(acc):
    iv = np.array([(0, 10, 20), (0, 5, 8)], cornetto_amd.IVL_DT)
    with pytest.raises(cornetto_amd.AccelError):
        acc.ivl_merge(iv, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("w,inc,dist,min_len", [(2500, 50, 1000, 30000), (300, 7, 0, 0), (1000, 1000, 5000, 3000)])
def test_cov_select_merged_equals_select_then_merge(acc, w, inc, dist, min_len):
    """the fused device path = cov_select, then the oracle's merge, then the length filter"""
    rng = np.random.default_rng(w)
    lens = [400_000, 90_000, 1_200_000, 5000]
    depths, mqs = [], []
    for ln in lens:
        d = rng.poisson(30, size=ln).astype(np.uint16)
        for _ in range(ln // 40000 + 1):
            a = int(rng.integers(0, ln)); b = min(ln, a + int(rng.integers(500, 60000)))
            d[a:b] = rng.choice([2, 120])
        q = (d * rng.uniform(0.2, 1.0)).astype(np.uint16)
        depths.append(d); mqs.append(q)
    cov = acc.cov_upload(depths, mqs)
    sums = acc.cov_prepare(cov, w, inc)
    mean = int(np.floor(sums[0] / sums[2] + 0.5))
    lo, hi = acc.cov_threshold(0.4, mean), acc.cov_threshold(2.5, mean)
    recs = acc.cov_select(cov, lo, hi, 0.4, 100000, 1000000, False)
    spans = np.zeros(len(recs), ob.SPAN_DT)
    spans["ctg"], spans["start"], spans["end"] = recs["ctg"], recs["st"], recs["end"]
    exp = ob.ivl_merge(spans, dist)
    exp = exp[exp["end"] - exp["start"] >= min_len]
    got = acc.cov_select_merged(cov, lo, hi, 0.4, 100000, 1000000, False, dist, min_len)
    cov.close()
    assert len(recs) > 100
    assert np.array_equal(_p2o(got), exp)


@pytest.mark.gpu
def test_noboringbits_panel_mode_cli(acc, tmp_path):
    """`noboringbits --panel`: steps 1-9 of scripts/create-cornetto.sh in one process = API select + oracle algebra"""
    import gzip
    import os
    import subprocess
    from helpers import read_bedgraph_pair
    golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    tot, mq = tmp_path / "tot.bg", tmp_path / "mq.bg"
    tot.write_bytes(gzip.open(os.path.join(golden_dir, "cov-total.bg.gz")).read())
    mq.write_bytes(gzip.open(os.path.join(golden_dir, "cov-mq20.bg.gz")).read())
    trip = read_bedgraph_pair(str(tot), str(mq))
    names, depths, mqs = [t[0] for t in trip], [t[1] for t in trip], [t[2] for t in trip]
    lens = [len(d) for d in depths]
    # assembly BED in a different order than the bedgraphs, with one contig the bedgraphs do not have
    order = list(range(len(names)))[::-1]
    asm_names = [names[i] for i in order] + [b"only_in_assembly"]
    asm_lens = [lens[i] for i in order] + [5000]
    (tmp_path / "asm.bed").write_bytes(b"".join(b"%s\t0\t%d\n" % (n, l) for n, l in zip(asm_names, asm_lens)))
    lowq_rows = [(asm_names[0], 100, 100 + 900), (asm_names[0], 5, 20), (b"unknown_ctg", 0, 5000), (asm_names[-1], 1000, 3000)]
    (tmp_path / "lowq.bed").write_bytes(b"".join(b"%s\t%d\t%d\tx\n" % r for r in lowq_rows))
    par = (300, 2000, 500, 700, 3000, 2500, 4000)
    p = subprocess.run([cornetto_amd.CLI_PATH, "noboringbits", str(tot), "-q", str(mq), "-w", "1000", "-i", "100", "-e", "2000", "-m", "10000",
                        "--panel", str(tmp_path / "asm.bed"), "--lowq", str(tmp_path / "lowq.bed"), "--panel-params", ",".join(map(str, par))],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr[-500:]
    # the same through the API and the oracle
    cov = acc.cov_upload(depths, mqs)
    sums = acc.cov_prepare(cov, 1000, 100)
    mean = int(np.floor(sums[0] / sums[2] + 0.5))
    lo, hi = acc.cov_threshold(0.4, mean), acc.cov_threshold(2.5, mean)
    recs = acc.cov_select(cov, lo, hi, 0.4, 2000, 10000, False)
    cov.close()
    spans = np.zeros(len(recs), ob.SPAN_DT)
    spans["ctg"], spans["start"], spans["end"] = recs["ctg"], recs["st"], recs["end"]
    fun = ob.ivl_merge(spans, par[0])
    fun = fun[fun["end"] - fun["start"] >= par[1]]
    idx = {n: i for i, n in enumerate(asm_names)}
    fun["ctg"] = [idx[names[c]] for c in fun["ctg"]]
    lowq = np.array([(idx[n], a, b) for n, a, b in lowq_rows if n in idx], ob.SPAN_DT)
    exp = ob.panel_boring(asm_lens, fun, lowq, par[2], par[3], par[4], par[5], par[6])
    text = b"".join(b"%s\t%d\t%d\n" % (asm_names[r["ctg"]], r["start"], r["end"]) for r in exp)
    assert len(exp) > 0
    assert p.stdout == text
