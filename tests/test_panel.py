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
