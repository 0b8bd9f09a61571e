"""GPU, at the bench workload's scale (the synthetic HG002-like assembly of bench.py).

  * 1 Gbp, EVERY contig against the CPU oracle (all four stages, record for record); the oracle runs on a thread per
    contig (ctypes releases the GIL);
  * sdust: canonical form and independence of the chunk decomposition;
  * a 4.5 Gbp layout whose contigs lie beyond byte offsets 2^31 and 2^32 (u16 coverage: element offsets beyond 2^31),
    oracle on the contigs that straddle those borders and on the last one;
  * a satellite-dense assembly (bench.py --profile satellite) against the oracle.
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_bind as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
THREADS = max(1, min(32, (os.cpu_count() or 2) - 1))


def _make(lens, seed, profile="uniform", coverage=True, dev=False):
    import torch
    from cornetto_amd import synth
    import cornetto_amd
    dev = torch.device("cuda", 0)
    bases, offs = synth.make_assembly(torch, dev, lens, seed, profile)
    depth = mq = None
    if coverage:
        depth, mq = synth.make_coverage(torch, dev, lens, offs, seed)
    torch.cuda.synchronize()
    acc = cornetto_amd.Accel(0, dev=dev)                  # dev: the development build of the library (the CORNETTO_SDUST_* switches exist there only)
    asm = acc.asm_wrap(bases.data_ptr(), offs, np.array(lens, dtype=np.int64))
    cov = acc.cov_wrap(depth.data_ptr(), mq.data_ptr(), offs, np.array(lens, dtype=np.int32)) if coverage else None
    return dict(torch=torch, acc=acc, asm=asm, cov=cov, lens=lens, offs=offs, bases=bases, depth=depth, mq=mq)


def _close(w):
    w["asm"].close()
    if w["cov"] is not None:
        w["cov"].close()
    w["acc"].close()


@pytest.fixture(scope="module")
def world():
    from cornetto_amd import synth
    w = _make(synth.contig_lengths(1_000_000_000), 7)
    yield w
    _close(w)


def _oracle_contig(w, ci, thr, stages):
    """the oracle's records of contig ci: dict stage -> array"""
    off, n = int(w["offs"][ci]), int(w["lens"][ci])
    out = {}
    seq = w["bases"][off:off + n].cpu().numpy()
    if "telo" in stages:
        oh = ob.telofind(seq, b"TTAGGG")
        out["hits"] = [(int(x["strand"]), int(x["start"]), int(x["end"])) for x in oh]
        out["wins"] = [(int(x["start"]), int(x["end"]), int(x["car"])) for x in ob.telowin(oh, n, thr)]
    if "sdust" in stages:
        out["sdust"] = np.asarray(ob.sdust(seq, 20, 64), dtype=np.uint64)
    if "cov" in stages:
        d = w["depth"][off:off + n].cpu().numpy().view(np.uint16)
        q = w["mq"][off:off + n].cpu().numpy().view(np.uint16)
        out["regs"] = ob.get_regs(d, q, 2500, 50)
    return out


def _compare_contigs(w, contigs, hits, wins, ivls, thr, stages=("telo", "sdust", "cov")):
    acc, cov = w["acc"], w["cov"]
    ob.lib()
    with ThreadPoolExecutor(THREADS) as ex:
        futs = {ci: ex.submit(_oracle_contig, w, ci, thr, stages) for ci in contigs}
        for ci in contigs:
            exp = futs[ci].result()
            if "telo" in stages:
                a, b = np.searchsorted(hits["ctg"], ci, "left"), np.searchsorted(hits["ctg"], ci, "right")
                assert [(int(x["strand"]), int(x["start"]), int(x["end"])) for x in hits[a:b]] == exp["hits"], ci
                a, b = np.searchsorted(wins["ctg"], ci, "left"), np.searchsorted(wins["ctg"], ci, "right")
                assert [(int(x["start"]), int(x["end"]), int(x["car"])) for x in wins[a:b]] == exp["wins"], ci
            if "sdust" in stages:
                a, b = np.searchsorted(ivls["ctg"], ci, "left"), np.searchsorted(ivls["ctg"], ci, "right")
                got = (ivls["start"][a:b].astype(np.uint64) << np.uint64(32)) | ivls["finish"][a:b].astype(np.uint32).astype(np.uint64)
                assert np.array_equal(got, exp["sdust"]), ci
            if "cov" in stages:
                got = acc.cov_regs(cov, ci)
                assert np.array_equal(got, exp["regs"].astype(got.dtype)), ci


def test_every_contig_equals_the_oracle_at_1gbp(world):
    acc, asm, cov, lens = world["acc"], world["asm"], world["cov"], world["lens"]
    thr = acc.telowin_threshold(0.4, 99.9)
    hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
    ivls = acc.sdust(asm, 20, 64)
    acc.cov_prepare(cov, 2500, 50)
    assert len(hits) > 100000 and len(ivls) > 10000
    _compare_contigs(world, list(range(len(lens))), hits, wins, ivls, thr)


@pytest.mark.timeout(1800)
def test_the_bench_assembly_itself_every_contig_at_3gbp():
    """the exact workload `python bench.py` times — 3 160 108 082 bases in 100 contigs, seed 0xC0FFEE, with its coverage — EVERY contig against the
    oracle: telomere runs and windows, sdust intervals, every coverage window (63 M).  bench.py itself checks 5 contigs (453 Mbases) of it against the
    reference's own functions in the timed run; this is the other 86 %."""
    from cornetto_amd import synth
    lens = synth.contig_lengths(0)
    assert sum(lens) == 3_160_108_082 and len(lens) == 100
    w = _make(lens, 0xC0FFEE)
    try:
        acc, asm, cov = w["acc"], w["asm"], w["cov"]
        thr = acc.telowin_threshold(0.4, 99.9)
        hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
        ivls = acc.sdust(asm, 20, 64)
        again = acc.sdust(asm, 20, 64)                      # (the second call: long chunks first, the one-go tail)
        assert np.array_equal(ivls, again)
        acc.cov_prepare(cov, 2500, 50)
        assert len(hits) > 1_000_000 and len(ivls) > 500_000
        _compare_contigs(w, list(range(len(lens))), hits, wins, ivls, thr)
    finally:
        _close(w)


@pytest.mark.timeout(3000)
@pytest.mark.parametrize("profile", ["humanlike", "satellite"])
def test_the_repeat_rich_bench_assemblies_every_contig_at_3gbp(profile):
    """bench.py's other two workload profiles at their full 3.16 Gbp: sdust intervals and telomere runs / windows of EVERY contig against the oracle (the
    stages the composition matters to: dp tiles, L2 skip, stepping, long chunks first).  About 2.4 minutes of 16 host cores for the two profiles (the
    oracle walks find_perfect at every base of the arrays); bench.py prints a number for each of these profiles, so the test runs always
    (CORNETTO_TEST_FULL=0 skips it on a small host)."""
    if os.environ.get("CORNETTO_TEST_FULL", "1") == "0":
        pytest.skip("CORNETTO_TEST_FULL=0")
    from cornetto_amd import synth
    lens = synth.contig_lengths(0)
    w = _make(lens, 0xC0FFEE, profile, coverage=False)
    try:
        acc, asm = w["acc"], w["asm"]
        thr = acc.telowin_threshold(0.4, 99.9)
        hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
        ivls = acc.sdust(asm, 20, 64)
        assert np.array_equal(ivls, acc.sdust(asm, 20, 64))
        assert len(ivls) > 900_000
        _compare_contigs(w, list(range(len(lens))), hits, wins, ivls, thr, stages=("telo", "sdust"))
    finally:
        _close(w)


def test_sdust_decomposition_invariance_and_canonical_form(world, monkeypatch):
    """the product library's intervals (its one compiled-in decomposition) equal the development build's under other chunk sizes, orders and
    wave counts over the same resident bases"""
    import cornetto_amd
    a = world["acc"].sdust(world["asm"], 20, 64).copy()
    acc = cornetto_amd.Accel(0, dev=True)
    asm = acc.asm_wrap(world["bases"].data_ptr(), world["offs"], np.array(world["lens"], dtype=np.int64))
    try:
        assert np.array_equal(a, acc.sdust(asm, 20, 64))
        monkeypatch.setenv("CORNETTO_SDUST_CHUNK", "4096")
        b = acc.sdust(asm, 20, 64).copy()
        monkeypatch.setenv("CORNETTO_SDUST_CHUNK", "777")
        monkeypatch.setenv("CORNETTO_SDUST_ORDER", "0")
        monkeypatch.setenv("CORNETTO_SDUST_WAVES", "1000")
        c = acc.sdust(asm, 20, 64).copy()
    finally:
        asm.close()
        acc.close()
    assert len(a) > 10000
    assert np.array_equal(a, b) and np.array_equal(a, c)
    # canonical: by contig, start ascending, disjoint and non-adjacent (src/sdust/sdust.c:94-98)
    same = a["ctg"][1:] == a["ctg"][:-1]
    assert np.all(np.diff(a["ctg"]) >= 0)
    assert np.all(a["start"][1:][same] > a["finish"][:-1][same])
    assert np.all(a["finish"] > a["start"])


def test_telofind_properties_and_planted_arrays(world):
    acc, asm, lens = world["acc"], world["asm"], world["lens"]
    thr = acc.telowin_threshold(0.4, 99.9)
    hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
    key = hits["ctg"].astype(np.int64) * 2 + hits["strand"]
    assert np.all(np.diff(key) >= 0)                                    # contig, then strand 0 before strand 1
    same = key[1:] == key[:-1]
    assert np.all(hits["start"][1:][same] > hits["end"][:-1][same])     # disjoint, and never touching (:57 resumes at end+1)
    assert np.all((hits["end"] - hits["start"]) % 6 == 0)
    for ci, n in enumerate(lens):
        if n < 50000:
            continue
        h = hits[hits["ctg"] == ci]
        rev = h[h["strand"] == 1]
        fwd = h[h["strand"] == 0]
        assert (0, 12000) in {(int(x["start"]), int(x["end"])) for x in rev}, ci          # CCCTAA x 2000 at the start
        assert (n - 9000, n) in {(int(x["start"]), int(x["end"])) for x in fwd}, ci       # TTAGGG x 1500 at the end
        w = wins[wins["ctg"] == ci]
        assert len(w) > 0 and int(w["start"][0]) == 0


def test_coverage_totals_and_selection(world):
    acc, cov, lens, torch = world["acc"], world["cov"], world["lens"], world["torch"]
    sd, sq, n = acc.cov_prepare(cov, 2500, 50)
    tot_d = tot_q = 0
    for off, ln in zip(world["offs"], lens):
        tot_d += int(world["depth"][int(off):int(off) + ln].to(torch.int64).sum().item())
        tot_q += int(world["mq"][int(off):int(off) + ln].to(torch.int64).sum().item())
    assert (sd, sq, n) == (tot_d, tot_q, sum(lens))
    mean = int(np.floor(sd / n + 0.5))
    lo, hi = acc.cov_threshold(0.4, mean), acc.cov_threshold(2.5, mean)
    recs = acc.cov_select(cov, lo, hi, 0.4, 100000, 1000000, False)
    assert len(recs) > 100000
    # the packed form of the same selection (8 B per window, contig by position)
    pk, first = acc.cov_select_packed(cov, lo, hi, 0.4, 100000, 1000000, False)
    assert len(first) == len(lens) + 1 and first[0] == 0 and first[-1] == len(pk) == len(recs)
    assert np.array_equal(acc.unpack_regs(pk, first, lens, 2500), recs)
    for boring in (True,):
        a = acc.cov_select(cov, lo, hi, 0.4, 100000, 1000000, boring)
        pk, first = acc.cov_select_packed(cov, lo, hi, 0.4, 100000, 1000000, boring)
        assert np.array_equal(acc.unpack_regs(pk, first, lens, 2500), a)
    key = recs["ctg"].astype(np.int64) * (1 << 32) + recs["st"]
    assert np.all(np.diff(key) > 0)                                      # print order: contig, then window
    # the selection = print_fun_bits' predicate (src/boringbits_main.c:439-440) over all windows of a contig
    for ci in (0, len(lens) // 2, len(lens) - 1):
        regs = acc.cov_regs(cov, ci)
        with np.errstate(divide="ignore", invalid="ignore"):
            flag = (regs["depth"] < lo) | (regs["depth"] > hi) | (regs["mq_depth"].astype(np.float64) / regs["depth"].astype(np.float64) < np.float64(np.float32(0.4)))
        exp = regs[flag] if lens[ci] >= 1000000 else regs[:0]
        got = recs[recs["ctg"] == ci]
        assert len(got) == len(exp), ci
        for k in ("st", "end", "depth", "mq_depth"):
            assert np.array_equal(got[k], exp[k]), (ci, k)


def test_cov_shard_sums_and_windows(world):
    """cornetto_cov_shard: two shares of the contigs on two handles; the totals add up to the whole and every contig's windows
    are the ones the whole object gives"""
    import cornetto_amd
    acc, cov, lens = world["acc"], world["cov"], world["lens"]
    whole = acc.cov_prepare(cov, 2500, 50)
    acc2 = cornetto_amd.Accel(0)
    try:
        a_idx = list(range(0, len(lens), 2))[::-1]           # any order
        b_idx = list(range(1, len(lens), 2))
        pa, pb = acc2.cov_shard(acc, cov, a_idx), acc.cov_shard(acc, cov, b_idx)
        sa, sb = acc2.cov_prepare(pa, 2500, 50), acc.cov_prepare(pb, 2500, 50)
        assert tuple(x + y for x, y in zip(sa, sb)) == whole
        acc.cov_prepare(cov, 2500, 50)
        for k in (0, len(a_idx) // 2, len(a_idx) - 1):
            assert np.array_equal(acc2.cov_regs(pa, k), acc.cov_regs(cov, a_idx[k]))
        pa.close()
        pb.close()
    finally:
        acc2.close()


def test_offsets_beyond_2_pow_32():
    """20 contigs of ~225 Mb = 4.5 Gbp: contig 9 straddles byte offset 2^31 of the bases, contig 19 byte offset 2^32;
    the u16 coverage arrays pass element offset 2^31 in contig 9 (byte offset 2^32) — the bench workload (3.16 Gbp) crosses
    the same borders"""
    lens = [225_000_000 + 1009 * i for i in range(20)]
    w = _make(lens, 3)
    try:
        offs = w["offs"]
        assert offs[9] < 2 ** 31 < offs[9] + lens[9] and offs[19] < 2 ** 32 < offs[19] + lens[19]
        acc, asm, cov = w["acc"], w["asm"], w["cov"]
        thr = acc.telowin_threshold(0.4, 99.9)
        hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
        ivls = acc.sdust(asm, 20, 64)
        sd, sq, n = acc.cov_prepare(cov, 2500, 50)
        assert n == sum(lens)
        _compare_contigs(w, [9, 10, 19], hits, wins, ivls, thr)
    finally:
        _close(w)
        del w


def test_satellite_dense_1gbp_through_the_sift_stages(monkeypatch):
    """bench.py --profile satellite at 1 Gbp, the kernel family chosen from the sample of the bases (sift / resolve: 3 % of them
    lie in satellite arrays): canonical result, two calls equal, and the four smallest of the contigs that hold arrays — each
    of 12 Mb or more, arrays of 0.1-5 Mb — record for record against the oracle"""
    from cornetto_amd import synth
    lens = synth.contig_lengths(1_000_000_000)
    w = _make(lens, 0xC0FFEE, "satellite", coverage=False)             # (the product build: sift / resolve by default)
    try:
        acc, asm = w["acc"], w["asm"]
        acc.set_timing(2)
        ivls = acc.sdust(asm, 20, 64)
        assert "sdust_prep" not in {n for n, _ in acc.last_timing()}          # (the per-lane kernel would have planned its queue)
        again = acc.sdust(asm, 20, 64)
        assert np.array_equal(ivls, again)
        masked = int((ivls["finish"].astype(np.int64) - ivls["start"]).sum())
        assert masked > 0.03 * sum(lens)
        c, st, fi = ivls["ctg"].astype(np.int64), ivls["start"].astype(np.int64), ivls["finish"].astype(np.int64)
        same = c[1:] == c[:-1]
        assert np.all(np.diff(c) >= 0) and np.all(st[1:][same] > fi[:-1][same]) and np.all(fi > st)
        big = sorted([i for i in range(len(lens)) if lens[i] >= 12_000_000], key=lambda i: lens[i])[:4]
        assert big
        thr = acc.telowin_threshold(0.4, 99.9)
        _compare_contigs(w, big, None, None, ivls, thr, stages=("sdust",))
    finally:
        _close(w)


def test_humanlike_1gbp_against_the_oracle():
    """bench.py --profile humanlike at 1 Gbp — isochores with 35-55 % GC, CpG at a fifth of its expectation, 10 % Alu-like and 15 %
    L1-like diverged copies, microsatellites, 3 % satellite arrays: the composition the sieve of sd_sift is sensitive to.  Canonical
    result, two calls equal, several times the masked fraction of uniform sequence outside the arrays, and the eight smallest of
    the contigs of 12 Mb or more record for record against the oracle (sdust and telofind / telowin)"""
    from cornetto_amd import synth
    lens = synth.contig_lengths(1_000_000_000)
    w = _make(lens, 0xC0FFEE, "humanlike", coverage=False)
    try:
        acc, asm = w["acc"], w["asm"]
        ivls = acc.sdust(asm, 20, 64)
        again = acc.sdust(asm, 20, 64)
        assert np.array_equal(ivls, again)
        c, st, fi = ivls["ctg"].astype(np.int64), ivls["start"].astype(np.int64), ivls["finish"].astype(np.int64)
        same = c[1:] == c[:-1]
        assert np.all(np.diff(c) >= 0) and np.all(st[1:][same] > fi[:-1][same]) and np.all(fi > st)
        masked = int((fi - st).sum())
        assert masked > 0.04 * sum(lens)                      # 3.2 % arrays + what the interspersed repeats and poly-A tails add
        small = [i for i in range(len(lens)) if 1_000_000 <= lens[i] < 12_000_000]          # no arrays there
        m_small = sum(int((fi - st)[c == i].sum()) for i in small) / float(sum(lens[i] for i in small))
        assert m_small > 0.008, m_small                       # (uniform sequence: 0.0016)
        big = sorted([i for i in range(len(lens)) if lens[i] >= 12_000_000], key=lambda i: lens[i])[:8]
        thr = acc.telowin_threshold(0.4, 99.9)
        hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
        _compare_contigs(w, big, hits, wins, ivls, thr, stages=("telo", "sdust"))
    finally:
        _close(w)


@pytest.mark.parametrize("dense", ["2", "0", "sift"])
def test_satellite_dense_assembly_equals_the_oracle(monkeypatch, dense):
    """bench.py --profile satellite at 30 Mb: (CATTC)n / (GGAAT)n arrays over > 3 % of the bases, microsatellites,
    poly-A runs — thousands of consecutive low-complexity chunks: through sdust_dense beside the main kernel (2; the default
    takes that route from 1024 such chunks on) and through the main kernel's queue alone (0)"""
    from cornetto_amd import synth
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", "1" if dense == "sift" else "0")
    monkeypatch.setenv("CORNETTO_SDUST_DENSE", "1" if dense == "sift" else dense)
    monkeypatch.setenv("CORNETTO_SDUST_DENSE_SPLIT", "3")     # the flagged chunks cut in three on the first call (off by default)
    lens = synth.contig_lengths(30_000_000)
    w = _make(lens, 5, "satellite", coverage=False, dev=True)
    try:
        acc, asm = w["acc"], w["asm"]
        thr = acc.telowin_threshold(0.4, 99.9)
        hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
        ivls = acc.sdust(asm, 20, 64)
        again = acc.sdust(asm, 20, 64)                      # the second call runs from the plan kept with the assembly
        assert np.array_equal(ivls, again)
        masked = int((ivls["finish"].astype(np.int64) - ivls["start"]).sum())
        assert masked > 0.03 * sum(lens)
        _compare_contigs(w, list(range(len(lens))), hits, wins, ivls, thr, stages=("telo", "sdust"))
    finally:
        _close(w)
