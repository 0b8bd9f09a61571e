"""CPU: the oracle restatement (oracle/oracle.c) against golden stdout of the unmodified reference
(tests/golden/*.exp, made by tests/golden/make_golden.py) and, where oracle/_ref is present, directly
against the reference's own functions through ctypes.  This is the parity PIN of the oracle."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_bind as ob
from helpers import (fmt_sdust, fmt_telofind, fmt_telowin, golden, read_bedgraph_pair, read_fastx)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _telofind_text(path, motif=b"TTAGGG"):
    out = []
    for name, _c, seq, _q in read_fastx(path):
        out.append(fmt_telofind(name, len(seq), ob.telofind(seq, motif)))
    return b"".join(out)


@pytest.mark.parametrize("fa,motif,exp", [
    ("probe.fa", b"TTAGGG", "probe.telofind.exp"),
    ("probe_selfoverlap.fa", b"AAAA", "probe_selfoverlap.AAAA.telofind.exp"),
    ("probe_selfoverlap.fa", b"ACAC", "probe_selfoverlap.ACAC.telofind.exp"),
    ("probe_selfoverlap.fa", b"ACACA", "probe_selfoverlap.ACACA.telofind.exp"),
    ("mix.fa.gz", b"TTAGGG", "mix.telofind.exp"),
    ("mix.fa.gz", b"ttaggg", "mix.lower_motif.telofind.exp"),
    ("mix.fa.gz", b"TTAGGGTTAGGG", "mix.k12.telofind.exp"),
    ("mix.fa.gz", b"TTAGGG" * 6, "mix.k36.telofind.exp"),
    ("mix.fa.gz", b"GGGTTA" * 11 + b"G", "mix.k67.telofind.exp"),
    ("mix.fa.gz", b"AAAA", "mix.AAAA.telofind.exp"),
    ("mix.fa.gz", b"GNG", "mix.GNG.telofind.exp"),
])
def test_telofind_golden(golden_dir, fa, motif, exp):
    assert _telofind_text(os.path.join(golden_dir, fa), motif) == golden(golden_dir, exp)


def _parse_telofind_tsv(data):
    """group consecutive lines by contig name exactly as src/telomere_windows.c:69-74 does"""
    groups = []
    for ln in data.splitlines():
        a = ln.split()
        if not groups or groups[-1][0] != a[0]:
            groups.append((a[0], int(a[1]), []))
        groups[-1][2].append((int(a[3]), int(a[4]), int(a[2])))
    return groups


def _telowin_text(tsv, identity, thr):
    t = ob.telowin_threshold(thr, identity)
    out = []
    for name, length, hs in _parse_telofind_tsv(tsv):
        hits = np.zeros(len(hs), dtype=ob.HIT_DT)
        for i, (s, e, st) in enumerate(hs):
            hits[i] = (s, e, st, 0)
        out.append(fmt_telowin(name, length, ob.telowin(hits, length, t)))
    return b"".join(out)


@pytest.mark.parametrize("tsv,identity,thr,exp", [
    ("probe.telomere", 99.9, 0.4, "probe.telowin.exp"),
    ("probe.telomere", 100.0, 0.5, "probe.i100t05.telowin.exp"),
    ("probe.telomere", 95.0, 0.4, "probe.i95.telowin.exp"),
    ("mix.telofind.exp", 99.9, 0.4, "mix.telowin.exp"),
    ("mix.telofind.exp", 99.9, 0.1, "mix.t01.telowin.exp"),
])
def test_telowin_golden(golden_dir, tsv, identity, thr, exp):
    assert _telowin_text(golden(golden_dir, tsv), identity, thr) == golden(golden_dir, exp)


def _sdust_text(path, T=20, W=64):
    return b"".join(fmt_sdust(name, ob.sdust(seq, T, W)) for name, _c, seq, _q in read_fastx(path))


@pytest.mark.parametrize("fa,T,W,exp", [
    ("probe.fa", 20, 64, "probe.sdust.exp"),
    ("probe_sdust.fa", 20, 64, "probe_sdust.sdust.exp"),
    ("probe_sdust.fa", 10, 32, "probe_sdust.w32t10.sdust.exp"),
    ("mix.fa.gz", 20, 64, "mix.sdust.exp"),
    ("mix.fa.gz", 10, 32, "mix.w32t10.sdust.exp"),
    ("mix.fa.gz", 25, 100, "mix.w100t25.sdust.exp"),
    ("mix.fa.gz", 30, 16, "mix.w16t30.sdust.exp"),
    ("mix.fa.gz", 5, 64, "mix.t5.sdust.exp"),
    ("reads.fq", 20, 64, "reads.sdust.exp"),
])
def test_sdust_golden(golden_dir, fa, T, W, exp):
    assert _sdust_text(os.path.join(golden_dir, fa), T, W) == golden(golden_dir, exp)


def panel_text(ctgs, boring, w=2500, inc=50, L=0.4, H=2.5, Q=0.4, m=1000000, e=100000):
    """the_boring_bits (src/boringbits_main.c:483-536) on oracle primitives"""
    tot = sum(float(d.astype(np.float64).sum()) for _n, d, _q in ctgs)
    totq = sum(float(q.astype(np.float64).sum()) for _n, _d, q in ctgs)
    n = sum(d.size for _n, d, _q in ctgs)
    mean = ob.mean_depth(tot, n)
    _ = ob.mean_depth(totq, n)
    lo, hi = ob.threshold(L, mean), ob.threshold(H, mean)
    out = []
    if any(ob.regs_assert(d.size, w, inc) for _, d, _ in ctgs):      # get_regs() over every contig comes first (:331-369): SIGABRT, no output
        return None
    for name, d, q in ctgs:
        regs = ob.get_regs(d, q, w, inc)
        length = d.size
        if boring:                                              # :463-481
            if length > m:
                for r in regs:
                    if r["st"] > e and r["end"] < length - e and not ob.is_fun(r["depth"], r["mq_depth"], lo, hi, Q):
                        out.append(b"%s\t%d\t%d\t%d\t%d\n" % (name, r["st"], r["end"], r["depth"], r["mq_depth"]))
        else:                                                   # :425-445
            if length < m:
                out.append(b"%s\t%d\t%d\t.\t.\n" % (name, 0, m))
            else:
                out.append(b"%s\t%d\t%d\t.\t.\n" % (name, 0, e))
                out.append(b"%s\t%d\t%d\t.\t.\n" % (name, length - e, length))
                for r in regs:
                    if ob.is_fun(r["depth"], r["mq_depth"], lo, hi, Q):
                        out.append(b"%s\t%d\t%d\t%d\t%d\n" % (name, r["st"], r["end"], r["depth"], r["mq_depth"]))
    return b"".join(out)


PANEL_CASES = [
    (True, dict(m=10000, e=1000, L=0.6, Q=0.6, H=1.6), "bg.boring_t1.exp"),
    (False, dict(H=2.5, L=0.5, Q=0.5, m=10000, e=1000), "bg.fun_t2.exp"),
    (False, dict(), "bg.fun_default.exp"),
    (True, dict(), "bg.boring_default.exp"),
    (False, dict(w=300, inc=7, L=0.33, H=1.45, Q=0.9, m=5000, e=500), "bg.fun_w300i7.exp"),
    (True, dict(w=300, inc=7, L=0.33, H=1.45, Q=0.9, m=5000, e=500), "bg.boring_w300i7.exp"),
    (False, dict(w=1000, inc=1000, m=2000, e=10000), "bg.fun_w1000i1000.exp"),
    (False, dict(w=300, inc=301, m=5000, e=500), "bg.fun_w300i301.exp"),
    (True, dict(w=300, inc=301, m=5000, e=500, L=0.33, H=1.45), "bg.boring_w300i301.exp"),
]
# the second pair of bedgraphs (sparse-*.bg.gz): -i larger than -w, -e beyond every contig, w % inc = 1
SPARSE_CASES = [
    (False, dict(w=64, inc=1000, m=1000, e=200), "sparse.fun_w64i1000.exp"),
    (True, dict(w=64, inc=1000, m=1000, e=200, L=0.2, H=3), "sparse.boring_w64i1000.exp"),
    (False, dict(w=64, inc=1000, m=100000), "sparse.fun_w64i1000_short.exp"),
    (False, dict(w=1982, inc=7, e=5000, m=1000), "sparse.fun_w1982i7e5000.exp"),
    (True, dict(w=1982, inc=7, e=5, m=1000, L=0.2, H=3), "sparse.boring_w1982i7.exp"),
    (False, dict(), "sparse.fun_default.exp"),
]
# (pair, sub-command, options) on which the reference raises SIGABRT with nothing printed (make_golden.py asserts the status)
ABORT_CASES = [
    ("cov", False, dict(w=300, inc=350)),
    ("cov", True, dict(w=64, inc=1000, m=100)),
    ("sparse", False, dict(w=64, inc=999)),
]


@pytest.fixture(scope="module")
def bg_ctgs(golden_dir):
    return read_bedgraph_pair(os.path.join(golden_dir, "cov-total.bg.gz"), os.path.join(golden_dir, "cov-mq20.bg.gz"))


@pytest.fixture(scope="module")
def sparse_ctgs(golden_dir):
    return read_bedgraph_pair(os.path.join(golden_dir, "sparse-total.bg.gz"), os.path.join(golden_dir, "sparse-mq20.bg.gz"))


@pytest.mark.parametrize("boring,kw,exp", PANEL_CASES)
def test_panel_golden(golden_dir, bg_ctgs, boring, kw, exp):
    assert panel_text(bg_ctgs, boring, **kw) == golden(golden_dir, exp)


@pytest.mark.parametrize("boring,kw,exp", SPARSE_CASES)
def test_panel_golden_sparse_windows(golden_dir, sparse_ctgs, boring, kw, exp):
    assert panel_text(sparse_ctgs, boring, **kw) == golden(golden_dir, exp)


@pytest.mark.parametrize("pair,boring,kw", ABORT_CASES)
def test_panel_where_the_reference_aborts(bg_ctgs, sparse_ctgs, pair, boring, kw):
    assert panel_text(bg_ctgs if pair == "cov" else sparse_ctgs, boring, **kw) is None


def test_regs_assert_closed_form_against_the_loop():
    """the product decides the asserts of get_regs() from the last window alone (cornetto_regs_assert); the oracle runs the reference's loop
    (src/boringbits_main.c:346-353,368-369).  Same answer on a grid that has every case: w < inc, w = inc, w > inc, len below / at / above w,
    lengths at and around multiples of inc."""
    import cornetto_amd
    L = cornetto_amd.lib()
    bad = 0
    for w in (1, 2, 7, 50, 64, 300, 2500):
        for inc in (1, 2, 7, 49, 50, 51, 64, 65, 299, 300, 301, 350, 1000, 2499, 2501):
            for length in list(range(1, 140)) + [299, 300, 301, 349, 350, 351, 700, 999, 1000, 1001, 1064, 1065, 2450, 2500, 2551, 5000, 15001, 15661]:
                a, b = L.cornetto_regs_assert(length, w, inc), ob.regs_assert(length, w, inc)
                assert (a != 0) == (b != 0) and (a == b or (a, b) == (353, 369)), (w, inc, length, a, b)
                bad += a != 0
                if inc <= w:
                    assert a == 0
    assert bad > 1000


def test_reference_exp_structure():
    """the reference ships expected outputs whose inputs were stripped: pin the structural facts
    they carry — 2500-wide windows, step 50, n_reg formula, edge lines first."""
    assert ob.n_reg(100000, 2500, 50) == 1951
    assert ob.n_reg(2551, 2500, 50) == 3
    assert ob.n_reg(120, 2500, 50) == 1
    assert ob.n_reg(2450, 2500, 50) == 1
    assert ob.n_reg(2401, 2500, 50) == 1
    # thresholds quoted in SURVEY appendix A-4 for mean 22
    assert (ob.threshold(0.6, 22), ob.threshold(1.6, 22)) == (13, 35)
    assert (ob.threshold(0.5, 22), ob.threshold(2.5, 22)) == (11, 55)


def bigenough_text(chroms, bed, T=50):
    """src/bigenough_main.c:229-296, :92-149, :152-227"""
    regs = {}
    for ln in chroms.splitlines():
        a = ln.split()
        regs[a[0]] = [int(a[1]), int(a[2]), 0]
    lines = [ln.split() for ln in bed.splitlines()]
    for a in lines:
        r = regs[a[0]]
        r[2] = int(np.int32(np.int64(r[2]) + np.int64(int(a[2]) - int(a[1]))))   # int covlen += int64
    out, csv = [], []
    for a in lines:
        r = regs[a[0]]
        if ob.bigenough_keep(r[2], r[0], r[1], T):
            out.append(b"%s\t%s\t%s\n" % (a[0], a[1], a[2]))
            csv.append(b"%s,%s,%s,+\n%s,%s,%s,-\n" % (a[0], a[1], a[2], a[0], a[1], a[2]))
    return b"".join(out), b"".join(csv)


@pytest.mark.parametrize("chroms,bed,T,exp_bed,exp_csv", [
    ("chroms.bed", "in.boringbits.bed", 50, "out.boringbits.bed", "out.boringbits.csv"),
    ("chroms.bed", "in_dip.boringbits.bed", 50, "out_dip.boringbits.bed", "out_dip.boringbits.csv"),
    ("ovf_chroms.bed", "ovf_in.bed", 50, "ovf_T50.bed.exp", "ovf_T50.csv.exp"),
    ("ovf_chroms.bed", "ovf_in.bed", 0, "ovf_T0.bed.exp", "ovf_T0.csv.exp"),
    ("ovf_chroms.bed", "ovf_in.bed", 100, "ovf_T100.bed.exp", "ovf_T100.csv.exp"),
    ("ovf_chroms.bed", "ovf_in.bed", 33, "ovf_T33.bed.exp", "ovf_T33.csv.exp"),
])
def test_bigenough_golden(golden_dir, chroms, bed, T, exp_bed, exp_csv):
    d = os.path.join(golden_dir, "bigenough")
    got_bed, got_csv = bigenough_text(golden(d, chroms), golden(d, bed), T)
    assert got_bed == golden(d, exp_bed)
    assert got_csv == golden(d, exp_csv)


# ---------------------------------------------------------------------------------------------------
# differential against the reference's own functions (only where oracle/_ref was built: this container)
# ---------------------------------------------------------------------------------------------------
REFSO = os.path.join(ROOT, "oracle", "_ref", "libcornetto_ref.so")


@pytest.mark.skipif(not os.path.exists(REFSO), reason="oracle/_ref not built (no /root/reference here)")
def test_sdust_vs_reference_function_random():
    ref = C.CDLL(REFSO)
    ref.sdust.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    ref.sdust.restype = C.POINTER(C.c_uint64)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b"ACGTacgtNNRY", dtype=np.uint8)
    for it in range(300):
        n = int(rng.integers(0, 3000))
        kind = it % 4
        if kind == 0:
            s = alpha[rng.integers(0, 4, size=n)]
        elif kind == 1:
            s = alpha[rng.integers(0, len(alpha), size=n)]
        elif kind == 2:   # STR mosaic
            parts = []
            while sum(map(len, parts)) < n:
                u = alpha[rng.integers(0, 4, size=int(rng.integers(1, 6)))]
                parts.append(np.tile(u, int(rng.integers(1, 40))))
                if rng.random() < 0.2:
                    parts.append(np.frombuffer(b"N" * int(rng.integers(1, 5)), dtype=np.uint8))
            s = np.concatenate(parts)[:n] if parts else np.zeros(0, np.uint8)
        else:             # two-letter low complexity
            s = alpha[rng.integers(0, 2, size=n)]
        s = np.ascontiguousarray(s, dtype=np.uint8)
        T = int(rng.choice([20, 20, 10, 5, 30]))
        W = int(rng.choice([64, 64, 32, 16, 100, 8]))
        cnt = C.c_int()
        buf = s.tobytes() + b"\0"
        r = ref.sdust(None, buf, len(s), T, W, C.byref(cnt))
        exp = np.array([r[i] for i in range(cnt.value)], dtype=np.uint64)
        libc.free(r)
        got = ob.sdust(s, T, W)
        assert np.array_equal(got, exp), (it, n, T, W)


def _bench_cpu_leg(kind, monkeypatch):
    import torch
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("CORNETTO_BENCH_BASELINE", kind)
    rng = np.random.default_rng(5)
    lens = [100000, 200000, 50000]
    offs = [0, 100032, 300032]
    n = 350100
    b = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    b[:600] = np.frombuffer(b"CCCTAA" * 100, dtype=np.uint8)
    b[100032 + 5000:100032 + 6200] = np.frombuffer(b"TTAGGG" * 200, dtype=np.uint8)
    b[100032 + 9000:100032 + 9300] = np.frombuffer(b"ttaggg" * 50, dtype=np.uint8)
    b[100032 + 20000:100032 + 20090] = np.frombuffer(b"AC" * 45, dtype=np.uint8)
    d = torch.from_numpy(rng.integers(0, 60, n).astype(np.int16))
    q = (d // 2).clone()
    base, results = bench.cpu_reference_leg(torch.from_numpy(b), d, q, offs, lens, [0, 1, 2], 310000)
    return bench, base, results, lens


def _as_gpu_records(results, min_len=60000):
    """the CPU leg's own results laid out like the four GPU record arrays (for the parity checker's self-test); a contig below
    min_len has no selected windows (src/boringbits_main.c:428)"""
    HIT = np.dtype([("ctg", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("end", "<i4")])
    WIN = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("end", "<i4"), ("car", "<i4")])
    IVL = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("finish", "<i4")])
    REC = np.dtype([("ctg", "<i4"), ("st", "<i4"), ("end", "<i4"), ("depth", "<i4"), ("mq_depth", "<i4")])
    hits, wins, ivls, recs = [], [], [], []
    lo, hi, q = 30, 30, 0.4
    for r in results:
        li = r["local"]
        hits += [(li, int(a), int(b), int(c)) for a, b, c in r["hits"]]
        for ln in r["wins_text"].splitlines():
            f = ln.split(b"\t")
            st, en = int(f[3]), int(f[4])
            marks = np.zeros(r["len"], np.uint8)
            for _s, a, b in r["hits"]:
                marks[a:b] = 1
            wins.append((li, st, en, int(marks[st:en].sum())))
        ivls += [(li, int(x) >> 32, int(x) & 0xFFFFFFFF) for x in r["sdust"]]
        for st, en, dp, mq in (r["regs"] if r["len"] >= min_len else []):
            flagged = dp < lo or dp > hi or (dp != 0 and mq / dp < np.float64(np.float32(q))) or (dp == 0 and False)
            if flagged:
                recs.append((li, st, en, dp, mq))
    return (np.array(hits, dtype=HIT), np.array(wins, dtype=WIN), np.array(ivls, dtype=IVL), np.array(recs, dtype=REC)), (lo, hi, q)


@pytest.mark.parametrize("kind", ["reference", "port"])
def test_bench_cpu_leg_and_parity_checker(capfd, monkeypatch, kind):
    """bench.py's CPU leg ("reference": find / process_scaffold / sdust / get_regs of the reference's shared object; "port":
    the oracle) keeps what it computes for the parity check; nothing those functions print may reach stdout (bench.py
    prints ONE JSON line there); the checker accepts matching records and names the first difference otherwise."""
    if kind == "reference" and not os.path.exists(REFSO):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    bench, base, results, lens = _bench_cpu_leg(kind, monkeypatch)
    assert base["kind"] == kind and base["cores"] == 1 and base["value"] > 0 and base["host_cores"] >= 1 and base["host_cpu"]
    assert set(base["stage_gbases_s"]) == {"telofind", "telowin", "sdust", "get_regs"}
    assert capfd.readouterr().out == ""
    # the whole leading contigs within the budget, and the smallest contig of all (below -m in the checker's call: the predicate that selects nothing)
    assert [r["local"] for r in results] == [0, 1, 2] and all(r["whole"] for r in results)
    assert len(results[1]["hits"]) >= 2 and len(results[1]["sdust"]) >= 1 and results[1]["wins_text"].count(b"\n") >= 1
    gpu, (lo, hi, q) = _as_gpu_records(results)
    par = bench.check_parity(results, gpu, lo, hi, q, 60000, lens)
    assert par["ok"], par
    assert par["checked_bases"] == 350000 and par["contigs"] == 3 and par["sdust_intervals"] == sum(len(r["sdust"]) for r in results)
    assert par["cov_windows_selected"] == len(gpu[3]) > 0
    # a single changed record of any stage is found
    for k in range(4):
        bad = [a.copy() for a in gpu]
        field = ("end", "car", "finish", "depth")[k]
        bad[k][field][-1] += 1
        assert not bench.check_parity(results, bad, lo, hi, q, 60000, lens)["ok"], k
    short = [gpu[0], gpu[1], gpu[2][:-1], gpu[3]]
    assert "sdust" in bench.check_parity(results, short, lo, hi, q, 60000, lens)["first_mismatch"]


def test_reference_and_port_cpu_legs_agree(monkeypatch):
    if not os.path.exists(REFSO):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    _b, _base, ref, _l = _bench_cpu_leg("reference", monkeypatch)
    _b, _base, port, _l = _bench_cpu_leg("port", monkeypatch)
    for a, b in zip(ref, port):
        assert np.array_equal(a["hits"], b["hits"]) and np.array_equal(a["sdust"], b["sdust"]) and np.array_equal(a["regs"], b["regs"])
        # the reference's process_scaffold runs with its built-in 0.4, the port with the adjusted threshold: a superset
        assert set(a["wins_text"].splitlines()) <= set(b["wins_text"].splitlines())


# ---- FASTA/FASTQ record framing ---------------------------------------------------
def _golden_text(golden_dir, f):
    import gzip
    t = open(os.path.join(golden_dir, f), "rb").read()
    return gzip.decompress(t) if f.endswith(".gz") else t


def test_fastx_framing_matches_reference_seq_and_fa2bed_goldens(golden_dir):
    from helpers import fmt_seq, fmt_fa2bed
    recs, rc = ob.fastx_parse(_golden_text(golden_dir, "reads.fq"))
    assert rc == -1 and len(recs) == 40
    assert fmt_seq(recs, 100) == _golden_text(golden_dir, "reads.m100.seq.exp")
    assert fmt_seq(recs, 30000) == _golden_text(golden_dir, "reads.default.seq.exp")
    for f, e in (("reads.fq", "reads.fa2bed.exp"), ("probe.fa", "probe.fa2bed.exp"), ("mix.fa.gz", "mix.fa2bed.exp")):
        recs, rc = ob.fastx_parse(_golden_text(golden_dir, f))
        assert fmt_fa2bed(recs) == _golden_text(golden_dir, e), f


def test_fastx_framing_status_codes():
    assert ob.fastx_parse(b"") == ([], -1)
    assert ob.fastx_parse(b"no header here\n") == ([], -1)
    assert ob.fastx_parse(b"@r1\nACGT\n+\nIII\n")[1] == -2                  # quality shorter than the read
    assert ob.fastx_parse(b"@r1\nACGT\n+") == ([], -2)                        # no quality at all
    recs, rc = ob.fastx_parse(b"@r1 c1\r\nAC\r\n+\r\nII\r\n>f x\nAC\nGT\n\n@r2\n\n+\n\n")
    assert rc == -1 and recs == [(b"r1", b"c1", b"AC", b"II"), (b"f", b"x", b"ACGT", None), (b"r2", b"", b"", b"")]


REFBIN = os.path.join(ROOT, "oracle", "_ref", "cornetto")


@pytest.mark.skipif(not os.path.exists(REFBIN), reason="oracle/_ref not built (no /root/reference here)")
def test_fastx_framing_vs_reference_binary_random(tmp_path):
    """`fa2bed` prints every record's name and length, `seq -m 0` the four strings; FASTA records and a read whose
    quality is cut off by the end of the file make `seq` print stale memory, so those texts are
    compared through fa2bed only."""
    import subprocess
    from helpers import fmt_seq, fmt_fa2bed, tricky_fastx
    rng = np.random.default_rng(3)
    f = str(tmp_path / "t.fq")
    n_seq = 0
    for it in range(150):
        text = tricky_fastx(rng, int(rng.integers(0, 12)), strict=(it % 3 == 0))
        open(f, "wb").write(text)
        recs, rc = ob.fastx_parse(text)
        assert subprocess.run([REFBIN, "fa2bed", f], capture_output=True).stdout == fmt_fa2bed(recs), it
        if all(q is not None for _, _, _, q in recs) and not text.endswith(b"+\n") and rc == -1:
            n_seq += 1
            assert subprocess.run([REFBIN, "seq", "-m", "0", f], capture_output=True).stdout == fmt_seq(recs, 0), it
    assert n_seq > 50


# ---- telobreaks -------------------------------------------------------------------
TELOBREAKS_CASES = [("mix.lens", "mix.sdust.exp", "mix.telofind.exp", "mix.breaks.exp"),
                    ("probe.lens", "probe.sdust.exp", "probe.telofind.exp", "probe.breaks.exp"),
                    ("tb_many.lens", "tb_many.sdust", "tb_many.telomere", "tb_many.breaks.exp")]


def telobreaks_text(golden_dir, lens_f, sd_f, tel_f, breaks_fn, order_fn):
    """the reference's stdout from array-level telobreaks + khash-order functions (oracle or product)"""
    from helpers import read_telobreaks_inputs, fmt_telobreaks
    names, lens, sd, tel = read_telobreaks_inputs(*(os.path.join(golden_dir, f) for f in (lens_f, sd_f, tel_f)))
    slot, order = order_fn(names)
    n_ids = int(slot.max()) + 1 if len(slot) else 0
    id_name, id_len = [None] * n_ids, [0] * n_ids
    for nm, ln, s in zip(names, lens, slot):
        if id_name[s] is None:
            id_name[s] = nm          # kh_put keeps the first key (src/khash.h:345)
        id_len[s] = ln               # the value is overwritten (src/telomere_breaks.c:69)
    ids = {nm: i for i, nm in enumerate(id_name)}
    sd_a = np.array([(ids[n], a, b) for n, a, b in sd if n in ids], dtype=ob.SPAN_DT)
    tel_a = np.array([(ids[n], a, b, m) for n, a, b, m in tel if n in ids], dtype=ob.TELROW_DT)
    res = breaks_fn(np.array(id_len, dtype=np.int32), sd_a, tel_a)
    out = []
    for cid in order:
        for r in res[res["ctg"] == cid]:
            out.append(fmt_telobreaks(id_name[cid], id_len[cid], int(r["start"]), int(r["end"])))
    return b"".join(out)


@pytest.mark.parametrize("lens_f,sd_f,tel_f,exp", TELOBREAKS_CASES)
def test_oracle_telobreaks_golden(golden_dir, lens_f, sd_f, tel_f, exp):
    assert telobreaks_text(golden_dir, lens_f, sd_f, tel_f, ob.telobreaks, ob.khash_order) == golden(golden_dir, exp)


def test_oracle_khash_order_growth():
    """bucket order across several table growths (4 -> 1024 buckets), names of mixed shapes"""
    names = [b"k%d" % (i * 7919 % 1000) for i in range(700)] + [b"k5", b"k12"]
    slot, order = ob.khash_order(names)
    assert len(order) == len(set(names)) and sorted(order) == list(range(len(order)))
    assert slot[700] == slot[names.index(b"k5")]
