"""GPU: cornetto_panel_step() — the other thread of a bench step as one call (cov_prepare -> thresholds -> cov_select_packed ->
telo_scan, queued in one go from the second step on, sized by the last step's counts and checked afterwards) gives exactly what the
four entry points give one after the other, whether its estimates hold or not."""
import os

import numpy as np
import pytest

import cornetto_amd
from helpers import read_bedgraph_pair, read_fastx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def acc():
    a = cornetto_amd.Accel(0)
    yield a
    a.close()


def _separate(acc, asm, cov, motif, thr, w, inc, L, H, Q, e, m, boring):
    sums = acc.cov_prepare(cov, w, inc)
    mean = int(np.floor(sums[0] / sums[2] + 0.5))
    lo, hi = acc.cov_threshold(L, mean), acc.cov_threshold(H, mean)
    recs, cf = acc.cov_select_packed(cov, lo, hi, Q, e, m, boring)
    hits, wins = acc.telo_scan(asm, motif, thr)
    return sums, (lo, hi), recs.copy(), cf.copy(), hits.copy(), wins.copy()


def _same(a, b):
    assert a[0] == b[0] and a[1] == b[1]
    for x, y in zip(a[2:], b[2:]):
        assert x.dtype == y.dtype and np.array_equal(x, y)


def _workload(rng, n_ctg=6, scale=1):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    lens = [int(x) for x in rng.integers(40_000 * scale, 160_000 * scale, size=n_ctg)] + [700, 64, 1]
    seqs, depths, mqs = [], [], []
    for n in lens:
        s = acgt[rng.integers(0, 4, size=n)].copy()
        for _ in range(max(1, n // 600)):                      # telomere units on both strands: thousands of runs, windows at the ends
            p = int(rng.integers(0, max(1, n - 200)))
            unit = b"TTAGGG" if rng.random() < 0.5 else b"CCCTAA"
            k = int(rng.integers(1, 40))
            rep = np.frombuffer(unit * k, dtype=np.uint8)
            s[p:p + len(rep)] = rep[:len(s[p:p + len(rep)])]
        if n > 3000:
            s[:1800] = np.frombuffer(b"CCCTAA" * 300, dtype=np.uint8)
            s[-1200:] = np.frombuffer(b"TTAGGG" * 200, dtype=np.uint8)
        seqs.append(s)
        d = rng.poisson(30, size=(n + 499) // 500).repeat(500)[:n].astype(np.uint16)
        q = np.minimum(d, rng.integers(0, 45, size=n)).astype(np.uint16)
        depths.append(d)
        mqs.append(q)
    return lens, seqs, depths, mqs


@pytest.mark.parametrize("lazy", [0, 1])
@pytest.mark.parametrize("boring", [False, True])
def test_panel_step_equals_the_four_calls(lazy, boring):
    rng = np.random.default_rng(2025 + lazy)
    lens, seqs, depths, mqs = _workload(rng)
    a = cornetto_amd.Accel(0)
    a.set_lazy(bool(lazy))
    asm = a.asm_upload(seqs)
    cov = a.cov_upload(depths, mqs)
    thr = a.telowin_threshold(0.4, 99.9)
    par = (b"TTAGGG", thr, 500, 50, 0.6, 1.4, 0.7, 1000, 5000, boring)
    ref = _separate(a, asm, cov, *par)
    a.wait()
    assert len(ref[2]) > 1000 and len(ref[4]) > 50 and len(ref[5]) > 5
    for i in range(4):                                          # 1st: no estimates for these objects yet (the exact calls above left some: either way the same)
        got = a.panel_step(asm, cov, *par)
        a.wait()
        _same(ref, got)
    # other parameters on the same objects: other keys, no estimate -> the exact entry points, then the queued form again
    par2 = (b"TTAGGG", a.telowin_threshold(0.2, 99.9), 2500, 50, 0.4, 2.5, 0.4, 2000, 50000, boring)
    ref2 = _separate(a, asm, cov, *par2)
    a.wait()
    for i in range(3):
        got = a.panel_step(asm, cov, *par2)
        a.wait()
        _same(ref2, got)
    got = a.panel_step(asm, cov, *par)
    a.wait()
    _same(ref, got)
    asm.close()
    cov.close()
    a.close()


@pytest.mark.parametrize("force", ["1", "64", "2000", "100000"])
def test_panel_step_with_estimates_that_do_not_hold(monkeypatch, force):
    """CORNETTO_STEP_EST_FORCE: lists, pairing and copies sized for `force` entries whatever the last step gave — where the counts are
    larger, nothing of the queued attempt is returned (no write beyond a list: tf_gather / tf_pair_dev check) and the exact entry
    points answer; the step after it is queued again"""
    rng = np.random.default_rng(77)
    lens, seqs, depths, mqs = _workload(rng, n_ctg=8, scale=2)
    a = cornetto_amd.Accel(0, dev=True)               # the development build: CORNETTO_STEP_EST_FORCE exists there only
    asm = a.asm_upload(seqs)
    cov = a.cov_upload(depths, mqs)
    par = (b"TTAGGG", a.telowin_threshold(0.4, 99.9), 500, 50, 0.6, 1.4, 0.7, 1000, 5000, False)
    ref = _separate(a, asm, cov, *par)
    assert len(ref[2]) > 2500 and len(ref[4]) > 1500
    _same(ref, a.panel_step(asm, cov, *par))
    monkeypatch.setenv("CORNETTO_STEP_EST_FORCE", force)
    _same(ref, a.panel_step(asm, cov, *par))
    _same(ref, a.panel_step(asm, cov, *par))
    monkeypatch.delenv("CORNETTO_STEP_EST_FORCE")
    _same(ref, a.panel_step(asm, cov, *par))
    _same(ref, a.panel_step(asm, cov, *par))
    asm.close()
    cov.close()
    a.close()


def test_panel_step_exchange_callback_and_errors(acc):
    """the exchange stands where the ranks all-reduce the three sums: what it returns decides the thresholds; an exception inside it
    comes back as that exception, with nothing leaked"""
    rng = np.random.default_rng(5)
    lens, seqs, depths, mqs = _workload(rng, n_ctg=3)
    asm = acc.asm_upload(seqs)
    cov = acc.cov_upload(depths, mqs)
    thr = acc.telowin_threshold(0.4, 99.9)
    seen = []

    def double(s):
        seen.append(tuple(s))
        return (2 * s[0], 2 * s[1], s[2])          # as if another rank of twice the depth had the same number of positions ... mean x 2
    sums, (lo, hi), recs, cf, hits, wins = acc.panel_step(asm, cov, b"TTAGGG", thr, 2500, 50, 0.4, 2.5, 0.4, 1000, 5000, False, double)
    own = acc.cov_prepare(cov, 2500, 50)
    assert seen == [own] and sums == (2 * own[0], 2 * own[1], own[2])
    mean = int(np.floor(sums[0] / sums[2] + 0.5))
    assert (lo, hi) == (acc.cov_threshold(0.4, mean), acc.cov_threshold(2.5, mean))
    r2, cf2 = acc.cov_select_packed(cov, lo, hi, 0.4, 1000, 5000, False)
    assert np.array_equal(recs, r2) and np.array_equal(cf, cf2)

    def boom(s):
        raise KeyError("rank 3 went away")
    with pytest.raises(KeyError):
        acc.panel_step(asm, cov, b"TTAGGG", thr, 2500, 50, 0.4, 2.5, 0.4, 1000, 5000, False, boom)
    got = acc.panel_step(asm, cov, b"TTAGGG", thr, 2500, 50, 0.4, 2.5, 0.4, 1000, 5000, False, double)      # the handle is still good
    assert np.array_equal(got[2], recs) and np.array_equal(got[4], hits)
    with pytest.raises(cornetto_amd.AccelError) as ei:                                                        # the asserts of get_regs() (-i > -w) pass through
        acc.panel_step(asm, cov, b"TTAGGG", thr, 64, 1000)
    assert ei.value.status == -7
    asm.close()
    cov.close()


def test_panel_step_bordered_motif_and_golden(acc, golden_dir):
    """a motif with a border (the sequential greedy rule: never queued) and the golden fixtures through the one call"""
    recs = read_fastx(os.path.join(golden_dir, "mix.fa.gz"))
    ctgs = read_bedgraph_pair(os.path.join(golden_dir, "cov-total.bg.gz"), os.path.join(golden_dir, "cov-mq20.bg.gz"))
    asm = acc.asm_upload([r[2] for r in recs])
    cov = acc.cov_upload([c[1] for c in ctgs], [c[2] for c in ctgs])
    thr = acc.telowin_threshold(0.4, 99.9)
    for motif in (b"TTAGGG", b"AAAA", b"TTAGGGTTAGGG"):
        par = (motif, thr, 300, 7, 0.33, 1.45, 0.9, 500, 5000, False)
        ref = _separate(acc, asm, cov, *par)
        for _ in range(3):
            _same(ref, acc.panel_step(asm, cov, *par))
    asm.close()
    cov.close()
