"""GPU parity tests: the HIP path, called through the C ABI (ctypes, include/cornetto_accel.h), against
(a) the golden stdout of the unmodified reference and (b) the CPU oracle on seeded inputs.
Bit-exact everywhere: integer / byte / index work."""
import os

import numpy as np
import pytest

import oracle_bind as ob
from helpers import (fmt_sdust, fmt_telofind, fmt_telowin, golden, read_bedgraph_pair, read_fastx)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def acc():
    import cornetto_amd
    a = cornetto_amd.Accel(0)
    yield a
    a.close()


def _records(golden_dir, fa):
    return read_fastx(os.path.join(golden_dir, fa))


def gpu_telofind_text(acc, recs, motif):
    asm = acc.asm_upload([r[2] for r in recs])
    hits = acc.telofind(asm, motif)
    asm.close()
    out = []
    for h in hits:
        name, ln = recs[h["ctg"]][0], len(recs[h["ctg"]][2])
        out.append(b"%s\t%d\t%d\t%d\t%d\t%d\n" % (name, ln, h["strand"], h["start"], h["end"], h["end"] - h["start"]))
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
def test_telofind_golden(acc, golden_dir, fa, motif, exp):
    assert gpu_telofind_text(acc, _records(golden_dir, fa), motif) == golden(golden_dir, exp)


def _parse_tsv(data):
    names, lens, hits = [], [], []
    for ln in data.splitlines():
        a = ln.split()
        if not names or names[-1] != a[0]:
            names.append(a[0])
            lens.append(int(a[1]))
        hits.append((len(names) - 1, int(a[2]), int(a[3]), int(a[4])))
    return names, lens, np.array(hits, dtype=[("ctg", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("end", "<i4")])


def _win_text(names, lens, wins):
    return b"".join(b"Window\t%s\t%d\t%d\t%d\t%s\n" % (names[w["ctg"]], lens[w["ctg"]], w["start"], w["end"],
                                                     ("%.3g" % (float(w["car"]) / float(w["end"] - w["start"]))).encode())
                    for w in wins)


@pytest.mark.parametrize("tsv,identity,thr,exp", [
    ("probe.telomere", 99.9, 0.4, "probe.telowin.exp"),
    ("probe.telomere", 100.0, 0.5, "probe.i100t05.telowin.exp"),
    ("probe.telomere", 95.0, 0.4, "probe.i95.telowin.exp"),
    ("mix.telofind.exp", 99.9, 0.4, "mix.telowin.exp"),
    ("mix.telofind.exp", 99.9, 0.1, "mix.t01.telowin.exp"),
])
def test_telowin_golden(acc, golden_dir, tsv, identity, thr, exp):
    names, lens, hits = _parse_tsv(golden(golden_dir, tsv))
    wins = acc.telowin(hits, lens, acc.telowin_threshold(thr, identity))
    assert _win_text(names, lens, wins) == golden(golden_dir, exp)


@pytest.mark.parametrize("motif", [b"TTAGGG", b"AAAA", b"CCCTAA", b"ACACA"])
def test_telo_scan_fused_vs_oracle(acc, golden_dir, motif):
    recs = _records(golden_dir, "mix.fa.gz")
    asm = acc.asm_upload([r[2] for r in recs])
    thr = acc.telowin_threshold(0.4, 99.9)
    hits, wins = acc.telo_scan(asm, motif, thr)
    asm.close()
    exp_h, exp_w = [], []
    for ci, r in enumerate(recs):
        oh = ob.telofind(r[2], motif)
        exp_h += [(ci, int(h["strand"]), int(h["start"]), int(h["end"])) for h in oh]
        exp_w += [(ci, int(w["start"]), int(w["end"]), int(w["car"])) for w in ob.telowin(oh, len(r[2]), thr)]
    assert [tuple(map(int, h)) for h in hits] == exp_h
    assert [tuple(map(int, w)) for w in wins] == exp_w


def gpu_sdust_text(acc, recs, T, W):
    asm = acc.asm_upload([r[2] for r in recs])
    iv = acc.sdust(asm, T, W)
    asm.close()
    return b"".join(b"%s\t%d\t%d\n" % (recs[x["ctg"]][0], x["start"], x["finish"]) for x in iv)


SDUST_CASES = [
    ("probe.fa", 20, 64, "probe.sdust.exp"),
    ("probe_sdust.fa", 20, 64, "probe_sdust.sdust.exp"),
    ("probe_sdust.fa", 10, 32, "probe_sdust.w32t10.sdust.exp"),
    ("mix.fa.gz", 20, 64, "mix.sdust.exp"),
    ("mix.fa.gz", 10, 32, "mix.w32t10.sdust.exp"),
    ("mix.fa.gz", 25, 100, "mix.w100t25.sdust.exp"),
    ("mix.fa.gz", 30, 16, "mix.w16t30.sdust.exp"),
    ("mix.fa.gz", 5, 64, "mix.t5.sdust.exp"),
    ("reads.fq", 20, 64, "reads.sdust.exp"),
]


@pytest.mark.parametrize("dense", ["2", "0", "sift"])
@pytest.mark.parametrize("chunk", ["0", "16", "100", "256", "1000", "2048", "4096"])
@pytest.mark.parametrize("fa,T,W,exp", SDUST_CASES)
def test_sdust_golden(dacc, golden_dir, monkeypatch, fa, T, W, exp, chunk, dense):
    """dense = "sift": the sift / resolve stages (chunk sizes they do not take fall back to the per-lane kernel by themselves);
    "2" / "0": the per-lane recurrence with / without its kernel for repeat arrays"""
    acc = dacc                                   # the development build: the switches below exist there only
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", "1" if dense == "sift" else "0")
    _sdust_golden(acc, golden_dir, monkeypatch, fa, T, W, exp, chunk, "1" if dense == "sift" else dense)


@pytest.mark.parametrize("fa,T,W,exp", SDUST_CASES)
def test_sdust_golden_product_build(acc, dacc, golden_dir, fa, T, W, exp):
    """the product build (no switches: its one compiled-in configuration) against the reference's stdout, and the development build without
    any switch set is the same configuration"""
    recs = _records(golden_dir, fa)
    assert gpu_sdust_text(acc, recs, T, W) == golden(golden_dir, exp)
    assert gpu_sdust_text(dacc, recs, T, W) == golden(golden_dir, exp)


def _sdust_golden(dacc, golden_dir, monkeypatch, fa, T, W, exp, chunk, dense):
    """chunk = bases per lane (0 = default heuristic); tiny chunks stress the speculative warm-up.  dense = 2: the chunks
    sampled as low-complexity always go to the per-lane kernel (sdust_dense) beside the main one (the default, 1, does that
    only when there are many of them), 0: everything to the main kernel"""
    acc = dacc                                   # the development build: the switches below exist there only
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", chunk)
    monkeypatch.setenv("CORNETTO_SDUST_DENSE", dense)
    assert gpu_sdust_text(acc, _records(golden_dir, fa), T, W) == golden(golden_dir, exp)


def _libc_free():
    import ctypes as C
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free.restype = None
    return libc.free


def test_sdust_dropin_signature(golden_dir):
    """cornetto_sdust(): same arguments and ownership as sdust() of src/sdust/sdust.h:19 (the caller free()s)"""
    import ctypes as C
    import cornetto_amd
    L = cornetto_amd.lib()
    free = _libc_free()
    recs = _records(golden_dir, "probe_sdust.fa")
    out = []
    for name, _c, seq, _q in recs:
        n = C.c_int()
        buf = C.create_string_buffer(seq, len(seq) + 1)
        r = L.cornetto_sdust(None, C.cast(buf, C.c_void_p), -1, 20, 64, C.byref(n))
        assert n.value >= 0
        out.append(fmt_sdust(name, np.array([r[i] for i in range(n.value)], dtype=np.uint64)))
        free(C.cast(r, C.c_void_p))
    assert b"".join(out) == golden(golden_dir, "probe_sdust.sdust.exp")


def _low_complexity_rich(n, seed):
    """n random bases with a short tandem repeat every ~700 bases: > 1 interval per kb"""
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    units = [b"A", b"AT", b"CATTC", b"AAAG", b"TTAGGG", b"GGAAT"]
    for k, p in enumerate(range(300, n - 400, 700)):
        u = units[k % len(units)]
        rep = np.frombuffer(u * (120 // len(u) + 1), dtype=np.uint8)[:60 + (k % 7) * 10]
        seq[p:p + len(rep)] = rep
    return seq


def _tiled_low_complexity(block, reps, seed):
    """`reps` copies of one low-complexity-rich block and the intervals the oracle gives for it.  sdust is local (its state
    is the last W-2 words), and the block borders lie in random sequence, so the intervals of every copy after the first
    are those of the second copy shifted — the reference itself needs ~1.6 us per base on such input (find_perfect walks P),
    the oracle is therefore run over two copies only (and that is checked on three)."""
    one = _low_complexity_rich(block, seed)
    two = ob.sdust(np.concatenate([one, one]), 20, 64)
    st, fi = (two >> np.uint64(32)).astype(np.int64), (two & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert not np.any((st < block) & (fi > block))                       # nothing straddles the border
    first = two[st < block]
    second_s, second_f = st[st >= block] - block, fi[st >= block] - block
    parts = [first]
    for k in range(1, reps):
        parts.append(((second_s + k * block).astype(np.uint64) << np.uint64(32)) | (second_f + k * block).astype(np.uint64))
    return np.tile(one, reps), np.concatenate(parts)


def test_tiled_expectation_holds_on_three_copies():
    seq, exp = _tiled_low_complexity(300_000, 3, 5)
    assert np.array_equal(ob.sdust(seq, 20, 64), exp)


def test_sdust_dropin_large_result_is_libc_freeable():
    """One 100 Mb low-complexity-rich sequence through the per-record call shape of src/sdust/sdust.c:199: the interval
    list (> 1 MiB: the library's pinned result pool inside) comes back as plain malloc memory the caller free()s, and
    equals the oracle's."""
    import ctypes as C
    import cornetto_amd
    L = cornetto_amd.lib()
    free = _libc_free()
    seq, exp = _tiled_low_complexity(2_000_000, 50, 11)
    n_bases = len(seq)
    assert len(exp) >= 90_000                      # >= 1 MiB of cornetto_ivl_t before the conversion
    for _ in range(2):                             # twice: the pool entry of the first call must have been returned
        n = C.c_int()
        r = L.cornetto_sdust(None, seq.ctypes.data_as(C.c_void_p), n_bases, 20, 64, C.byref(n))
        assert n.value == len(exp)
        got = np.ctypeslib.as_array(r, shape=(n.value,)).copy()
        free(C.cast(r, C.c_void_p))
        assert np.array_equal(got, exp)


def test_sdust_core_buffered_interface(golden_dir):
    """cornetto_sdust_buf_init / cornetto_sdust_core / cornetto_sdust_buf_destroy: src/sdust/sdust.h:16-21 — the result
    belongs to the buf and stays valid until the next call on it"""
    import ctypes as C
    import cornetto_amd
    L = cornetto_amd.lib()
    recs = _records(golden_dir, "mix.fa.gz")
    buf = L.cornetto_sdust_buf_init(None)
    assert buf
    assert not L.cornetto_sdust_buf_init(C.c_void_p(1))          # kalloc pools: not supported, like km != NULL in cornetto_sdust
    out = []
    for name, _c, seq, _q in recs:
        n = C.c_int()
        arr = np.frombuffer(seq, dtype=np.uint8)
        r = L.cornetto_sdust_core(arr.ctypes.data_as(C.c_void_p), len(seq), 20, 64, C.byref(n), buf)
        assert n.value >= 0 and r
        out.append(fmt_sdust(name, np.array([r[i] for i in range(n.value)], dtype=np.uint64)))
    L.cornetto_sdust_buf_destroy(buf)
    assert b"".join(out) == golden(golden_dir, "mix.sdust.exp")


def _rand_seqs(rng, n_seq, kind):
    alpha = np.frombuffer(b"ACGTacgtNNRY", dtype=np.uint8)
    seqs = []
    for it in range(n_seq):
        n = int(rng.integers(0, 6000))
        k = kind if kind >= 0 else it % 4
        if k == 0:
            s = alpha[rng.integers(0, 4, size=n)]
        elif k == 1:
            s = alpha[rng.integers(0, len(alpha), size=n)]
        elif k == 2:
            parts = []
            while sum(map(len, parts)) < n:
                u = alpha[rng.integers(0, 4, size=int(rng.integers(1, 6)))]
                parts.append(np.tile(u, int(rng.integers(1, 40))))
                if rng.random() < 0.2:
                    parts.append(np.frombuffer(b"N" * int(rng.integers(1, 5)), dtype=np.uint8))
            s = np.concatenate(parts)[:n] if parts else np.zeros(0, np.uint8)
        else:
            s = alpha[rng.integers(0, 2, size=n)]
        seqs.append(np.ascontiguousarray(s, dtype=np.uint8))
    return seqs


@pytest.mark.parametrize("T,W,chunk", [(20, 64, "37"), (20, 64, "512"), (10, 32, "64"), (5, 64, "200"), (30, 16, "16"),
                                       (25, 100, "300"), (20, 8, "50"), (20, 257, "700"), (1, 3, "40"),
                                       (0, 64, "90"), (4, 64, "300"), (3, 20, "64"), (9, 64, "256"), (100, 64, "256")])
def test_sdust_random_vs_oracle(dacc, monkeypatch, T, W, chunk):
    acc = dacc                                   # the development build: the switches below exist there only
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", chunk)
    # a small grid: every lane takes many chunks from the queue, of unequal lengths, one after the other
    monkeypatch.setenv("CORNETTO_SDUST_WAVES", str(1 + (T + W) % 3))
    rng = np.random.default_rng(1234 + T * 1000 + W)
    seqs = _rand_seqs(rng, 120, -1)
    asm = acc.asm_upload(seqs)
    iv = acc.sdust(asm, T, W)
    asm.close()
    got = [(int(x["ctg"]), int(x["start"]), int(x["finish"])) for x in iv]
    exp = []
    for ci, s in enumerate(seqs):
        for r in ob.sdust(s, T, W):
            r = int(r)
            exp.append((ci, r >> 32, r & 0xFFFFFFFF))
    assert got == exp


@pytest.mark.parametrize("seed", list(range(48)))
def test_sdust_random_thresholds_vs_oracle(dacc, monkeypatch, seed):
    """thresholds and windows drawn at random (T 5..120, W 3..66: m = T / 5 from 1 to 24, the bounds of the pass trigger change with
    every pair), chunk sizes down to a few windows, on random / tandem / two-letter sequences with non-bases"""
    acc = dacc                                   # the development build: the switches below exist there only
    rng = np.random.default_rng(9000 + seed)
    T = int(rng.integers(5, 121))
    W = int(rng.integers(3, 67))
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", str(int(rng.integers(max(16, W), 900))))
    monkeypatch.setenv("CORNETTO_SDUST_WAVES", str(1 + seed % 3))
    if seed % 2:
        monkeypatch.setenv("CORNETTO_SDUST_DENSE", "2")     # every chunk sampled as low-complexity goes to the per-lane kernel sdust_dense
    seqs = _rand_seqs(rng, 40, -1)
    if seed % 2:                                            # ... and there are such chunks: tandem arrays longer than a chunk
        u = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(1, 7)))]
        seqs.append(np.tile(u, 6000 // len(u)))
        seqs.append(np.concatenate([seqs[0][:500], np.tile(np.frombuffer(b"CATTC", dtype=np.uint8), 700), seqs[0][:300]]))
    asm = acc.asm_upload(seqs)
    iv = acc.sdust(asm, T, W)
    asm.close()
    got = [(int(x["ctg"]), int(x["start"]), int(x["finish"])) for x in iv]
    exp = []
    for ci, q in enumerate(seqs):
        for r in ob.sdust(q, T, W):
            r = int(r)
            exp.append((ci, r >> 32, r & 0xFFFFFFFF))
    assert got == exp, (T, W)


def _sift_stress_seq(rng, n, kind):
    """n bases for the sift / resolve stages: random sequence with tandem repeats of every period, satellite arrays (exact and
    with substitutions), homopolymers, and — kind 1 — N runs, lower case and bytes that are not letters, which send the chunks
    around them to the sequential kernel"""
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    for _ in range(n // 1500):
        p = int(rng.integers(0, n))
        u = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(1, 8)))]
        L = int(rng.integers(8, 400 if rng.random() < 0.3 else 60))
        rep = np.tile(u, L // len(u) + 1)[:L].copy()
        sub = rng.random(L) < 0.02
        rep[sub] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(sub.sum()))]
        s[p:p + L] = rep[:len(s[p:p + L])]
    if n > 20000:
        p, L = n // 3, n // 6
        arr = np.tile(np.frombuffer(b"CATTC", dtype=np.uint8), L // 5 + 1)[:L].copy()
        sub = rng.random(L) < 0.02
        arr[sub] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(sub.sum()))]
        s[p:p + L] = arr
        s[n - 3000:n - 1000] = np.tile(np.frombuffer(b"GGAAT", dtype=np.uint8), 400)      # exact, up to near the end
    if kind == 1 and n >= 3:
        for _ in range(max(1, n // 20000)):
            p = int(rng.integers(0, n))
            L = int(rng.integers(1, 300))
            s[p:p + L] = ord("N")
        for _ in range(max(1, n // 30000)):
            p = int(rng.integers(0, n))
            s[p:p + int(rng.integers(1, 200))] |= 0x20
        for _ in range(max(1, n // 50000)):
            s[int(rng.integers(0, n))] = int(rng.choice(np.frombuffer(b"RYKM-*\x01\x03", dtype=np.uint8)))
    return s


@pytest.mark.parametrize("T,W,chunk,kind", [
    (20, 64, "0", 0), (20, 64, "0", 1), (20, 64, "256", 0), (20, 64, "256", 1), (20, 64, "320", 1), (20, 64, "1024", 1),
    (20, 64, "3968", 1), (10, 32, "256", 1), (10, 32, "0", 0), (5, 64, "512", 1), (30, 16, "256", 1), (25, 66, "448", 1),
    (7, 20, "256", 0), (50, 50, "1792", 1), (100, 64, "0", 1), (12, 3, "256", 1), (20, 4, "256", 1), (20, 65, "640", 1)])
def test_sdust_sift_vs_oracle(dacc, monkeypatch, T, W, chunk, kind):
    """the sift / resolve stages (sdust_sift.hpp) on sequences of many chunks: chunk borders inside repeat arrays, contigs that
    end inside an array, contigs shorter than a tile, chunks handed to the sequential kernel beside chunks that are not"""
    acc = dacc                                   # the development build: the switches below exist there only
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", chunk)
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", "1")
    rng = np.random.default_rng(4242 + T * 131 + W + int(chunk) + kind)
    seqs = [_sift_stress_seq(rng, n, kind) for n in (150_000, 70_001, 4097, 1792, 1793, 257, 256, 255, 130, 65, 64, 63, 5, 3, 2, 1, 0, 30_000)]
    seqs.append(np.tile(np.frombuffer(b"A", dtype=np.uint8), 5000))
    seqs.append(np.tile(np.frombuffer(b"AC", dtype=np.uint8), 2500))
    asm = acc.asm_upload(seqs)
    iv = acc.sdust(asm, T, W)
    asm.close()
    got = [(int(x["ctg"]), int(x["start"]), int(x["finish"])) for x in iv]
    exp = []
    for ci, q in enumerate(seqs):
        for r in ob.sdust(q, T, W):
            r = int(r)
            exp.append((ci, r >> 32, r & 0xFFFFFFFF))
    assert got == exp, (T, W, chunk)


@pytest.mark.parametrize("dp,l2skip", [("1", "65"), ("1", "1"), ("24", "48"), ("65", "1"), ("65", "65"), ("8", "32")])
@pytest.mark.parametrize("T,W,chunk", [(20, 64, "0"), (20, 66, "448"), (20, 65, "640"), (12, 40, "256"), (5, 7, "256"), (30, 16, "1024"), (2, 64, "0")])
def test_sdust_dp_tiles_and_l2_skip_vs_oracle(dacc, monkeypatch, T, W, chunk, dp, l2skip):
    """round 4: tiles resolved end-parallel (CORNETTO_SIFT_DP: from how many sifted positions per tile — 1: every tile that
    holds one, 65: none) and L1 batches that skip the second filter (CORNETTO_SIFT_L2SKIP) are scheduling choices: any
    superset of the inserting positions and either resolve stage give the reference's intervals.  Exact and diverged
    repeats of periods 1-7, chunk borders and contig ends inside them, hand-overs stepping -> dp -> dp -> stepping."""
    acc = dacc                                   # the development build: the switches below exist there only
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", chunk)
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", "1")
    monkeypatch.setenv("CORNETTO_SIFT_DP", dp)
    monkeypatch.setenv("CORNETTO_SIFT_L2SKIP", l2skip)
    rng = np.random.default_rng(99 + T * 7 + W)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for k, n in enumerate((60_000, 20_011, 4096, 1537, 600, 129)):
        s = acgt[rng.integers(0, 4, size=n)].copy()
        for _ in range(max(1, n // 2000)):
            L = int(rng.integers(8, min(1500, n)))
            p = int(rng.integers(0, n - L + 1))
            u = int(rng.integers(1, 8))
            rep = np.tile(acgt[rng.integers(0, 4, size=u)], L // u + 1)[:L].copy()
            if k % 3:
                mm = rng.random(L) < 0.02 * (k % 3)
                rep[mm] = acgt[rng.integers(0, 4, size=int(mm.sum()))]
            s[p:p + L] = rep
        if k == 1:
            s[-700:] = np.tile(np.frombuffer(b"TTAGGG", dtype=np.uint8), 117)[:700]      # the contig ends inside an array
        if k == 2:
            s[1000:1003] = ord("N")                                                      # one walked chunk among the others
        seqs.append(s)
    asm = acc.asm_upload(seqs)
    iv = acc.sdust(asm, T, W)
    asm.close()
    got = [(int(x["ctg"]), int(x["start"]), int(x["finish"])) for x in iv]
    exp = [(ci, int(r) >> 32, int(r) & 0xFFFFFFFF) for ci, q in enumerate(seqs) for r in ob.sdust(q, T, W)]
    assert got == exp, (T, W, chunk, dp, l2skip)


def test_sdust_sift_off_equals_on(dacc, monkeypatch):
    """CORNETTO_SDUST_SIFT=0 (the per-lane recurrence of sdust_w64 for every chunk) and the default give the same intervals"""
    acc = dacc                                   # the development build: the switches below exist there only
    rng = np.random.default_rng(77)
    seqs = [_sift_stress_seq(rng, 400_000, 1), _sift_stress_seq(rng, 90_000, 0)]
    asm = acc.asm_upload(seqs)
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", "1")
    on = acc.sdust(asm, 20, 64)
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", "0")
    off = acc.sdust(asm, 20, 64)
    asm.close()
    assert len(on) > 200 and np.array_equal(on, off)


def test_sdust_fused_tail_with_an_estimate_that_does_not_hold(dacc, monkeypatch):
    """the one-launch tail of a repeated call (gather + st_fused + copy, sized by the last call's counts) when the counts of THIS call
    are larger than the estimate: no tile of st_fused may touch anything (rows beyond the estimate were never gathered, their heads
    would land behind the output block) and the call takes the long way — same intervals, on a fresh and on the grown workspace"""
    acc = dacc                                   # the development build: the switches below exist there only
    rng = np.random.default_rng(515)
    seqs = [_sift_stress_seq(rng, 2_500_000, 1), _sift_stress_seq(rng, 90_000, 0)]
    asm = acc.asm_upload(seqs)
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", "1")
    first = acc.sdust(asm, 20, 64)
    again = acc.sdust(asm, 20, 64)                       # the fused tail, estimate from the first call
    assert len(first) > 1200 and np.array_equal(first, again)
    for force in ("1", "1000", "1025", str(len(first) // 2)):
        monkeypatch.setenv("CORNETTO_SDUST_EST_FORCE", force)
        short = acc.sdust(asm, 20, 64)                   # estimate too small: st_fused returns at once, the long way answers
        monkeypatch.delenv("CORNETTO_SDUST_EST_FORCE")
        assert np.array_equal(first, short), force
        assert np.array_equal(first, acc.sdust(asm, 20, 64))     # and the estimate is rebuilt (the long way leaves its counts)
        assert np.array_equal(first, acc.sdust(asm, 20, 64))
    asm.close()


def test_sdust_boost_from_another_thread_does_not_change_results(dacc, monkeypatch):
    """cornetto_accel_set_share + cornetto_accel_boost: the resident sift waves take part of the chip; another host thread says "the rest is
    free now" while the call runs and the remaining waves are launched as a second kernel on the same chunk counters — same intervals,
    whenever the flag arrives (before the call, early, late, never)"""
    acc = dacc                                   # the development build: the switches below exist there only
    import threading
    import time
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", "1")
    monkeypatch.delenv("CORNETTO_SDUST_CHUNK", raising=False)
    rng = np.random.default_rng(31)
    seqs = [_sift_stress_seq(rng, 20_000_000, 1), _sift_stress_seq(rng, 3_000_000, 0)]
    asm = acc.asm_upload(seqs)
    try:
        ref = acc.sdust(asm, 20, 64)
        acc.set_share(40)
        for delay in (None, 0.0, 0.0005, 0.003, -1.0):
            acc.boost(delay is not None and delay < 0)              # -1: on before the call starts
            t = None
            if delay is not None and delay >= 0:
                t = threading.Thread(target=lambda d=delay: (time.sleep(d), acc.boost(True)))
                t.start()
            got = acc.sdust(asm, 20, 64)
            if t:
                t.join()
            assert np.array_equal(got, ref), delay
    finally:
        acc.boost(False)
        acc.set_share(100)
        asm.close()
    assert len(ref) > 5000


def test_sdust_family_choice_by_sample_keeps_small_assemblies_on_the_sift_stages(dacc, monkeypatch):
    """CORNETTO_SDUST_SIFT=-1 (the choice by a sample of the bases): an assembly below 2 Gbases takes the sift stages whatever its
    composition (the per-lane kernel has a 4 ms floor); the intervals are the oracle's, plain or repeat-rich, call after call"""
    acc = dacc                                   # the development build: the switches below exist there only
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", "-1")
    monkeypatch.delenv("CORNETTO_SDUST_CHUNK", raising=False)
    rng = np.random.default_rng(5)
    plain = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=1_000_000)].copy()
    rich = plain.copy()
    rich[200_000:400_000] = np.tile(np.frombuffer(b"CATTC", dtype=np.uint8), 40_000)
    rich[600_000:600_300] = ord("N")
    acc.set_timing(2)
    try:
        for seq in (plain, rich):
            asm = acc.asm_upload([seq])
            iv = acc.sdust(asm, 20, 64)
            names = {n for n, _ in acc.last_timing()}
            assert "sdust_prep" not in names, names                    # (only the per-lane kernel plans its queue)
            st = acc.sdust_stats(asm, 20, 64)
            assert st is None or st.get("kernel") == "sd_sift", st
            iv2 = acc.sdust(asm, 20, 64)
            asm.close()
            assert np.array_equal(iv, iv2)
            assert [(int(x["start"]), int(x["finish"])) for x in iv] == [(int(r) >> 32, int(r) & 0xFFFFFFFF) for r in ob.sdust(seq, 20, 64)]
    finally:
        acc.set_timing(0)


def test_sdust_largest_window_on_homopolymers(acc):
    """W = 257: the window holds 255 words and a homopolymer fills it with 255 copies of one 3-mer — the most the byte
    counters of the older kernel hold; from W = 258 on another kernel (32-bit counters, state in global memory) takes over"""
    import cornetto_amd
    rng = np.random.default_rng(3)
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=6000)].copy()
    s[1000:1700] = ord("A")
    s[3000:3300] = np.frombuffer(b"AC" * 150, dtype=np.uint8)
    s[4000:4258] = ord("T")
    asm = acc.asm_upload([s])
    for T in (20, 100):
        iv = acc.sdust(asm, T, 257)
        assert [(int(x["start"]), int(x["finish"])) for x in iv] == [(int(r) >> 32, int(r) & 0xFFFFFFFF) for r in ob.sdust(s, T, 257)]
    # wider windows: the kernel that keeps its state in global memory (258 <= W <= 1026); small inputs: it walks a window of up
    # to 1024 words out of global memory at every candidate step
    big = np.concatenate([s[900:1150], s[3000:3100], np.frombuffer(b"ACG" * 60, dtype=np.uint8), s[:120]])
    asm2 = acc.asm_upload([big, s[:50], np.zeros(0, np.uint8)])
    for T, W in ((20, 258), (35, 530), (20, 1026)):
        iv = acc.sdust(asm2, T, W)
        exp = [(ci, int(r) >> 32, int(r) & 0xFFFFFFFF) for ci, q in enumerate((big, s[:50], np.zeros(0, np.uint8))) for r in ob.sdust(q, T, W)]
        assert [(int(x["ctg"]), int(x["start"]), int(x["finish"])) for x in iv] == exp, (T, W)
    with pytest.raises(cornetto_amd.AccelError):
        acc.sdust(asm, 20, 1027)
    asm2.close()
    asm.close()


def test_telo_scan_repeated_on_one_handle_with_changing_assemblies(acc):
    """the mark bitmap block of a handle keeps the zero padding of the last assembly's layout from one call to the next (no fill
    in front of the second scan of the same assembly): alternate two layouts, change a contig's bytes in place between calls, change
    the motif — the windows and the runs follow the oracle every time"""
    rng = np.random.default_rng(404)
    thr = acc.telowin_threshold(0.4, 99.9)

    def make(n_seq, top):
        seqs = []
        for _ in range(n_seq):
            n = int(rng.integers(1, top))
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
            for motif in (b"TTAGGG", b"CCCTAA", b"AAAA"):
                if n > 4000:
                    p = int(rng.integers(0, n - 3000))
                    rep = np.frombuffer(motif * int(rng.integers(50, 400)), dtype=np.uint8)[: n - p]
                    s[p:p + len(rep)] = rep
            seqs.append(s)
        return seqs

    def check(asm, seqs, motif):
        hits, wins = acc.telo_scan(asm, motif, thr)
        exp_h, exp_w = [], []
        for ci, s in enumerate(seqs):
            oh = ob.telofind(s, motif)
            exp_h += [(ci, int(h["strand"]), int(h["start"]), int(h["end"])) for h in oh]
            exp_w += [(ci, int(w["start"]), int(w["end"]), int(w["car"])) for w in ob.telowin(oh, len(s), thr)]
        assert [tuple(map(int, h)) for h in hits] == exp_h, motif
        assert [tuple(map(int, w)) for w in wins] == exp_w, motif

    big, small = make(12, 60000), make(30, 9000)
    a_big, a_small = acc.asm_upload(big), acc.asm_upload(small)
    for asm, seqs, motif in ((a_big, big, b"TTAGGG"), (a_big, big, b"TTAGGG"), (a_big, big, b"AAAA"), (a_small, small, b"TTAGGG"), (a_small, small, b"CCCTAA"),
                             (a_big, big, b"CCCTAA"), (a_big, big, b"TTAGGG"), (a_small, small, b"TTAGGG"), (a_small, small, b"TTAGGG")):
        check(asm, seqs, motif)
    a_small.close()
    # the same resident object, other bytes: fewer marks than the call before left in the block
    import torch
    flat = torch.zeros(sum((len(s) + 63) // 64 * 64 for s in big), dtype=torch.uint8, device="cuda")
    offs, o = [], 0
    for s in big:
        offs.append(o)
        flat[o:o + len(s)] = torch.from_numpy(s).cuda()
        o += (len(s) + 63) // 64 * 64
    w = acc.asm_wrap(flat.data_ptr(), np.array(offs, dtype=np.int64), np.array([len(s) for s in big], dtype=np.int64))
    check(w, big, b"TTAGGG")
    check(w, big, b"TTAGGG")
    plain = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=len(s))].copy() for s in big]
    for s, o in zip(plain, offs):
        flat[o:o + len(s)] = torch.from_numpy(s).cuda()
    torch.cuda.synchronize()
    check(w, plain, b"TTAGGG")
    w.close()
    a_big.close()


def test_telofind_random_vs_oracle(acc):
    rng = np.random.default_rng(99)
    for motif in (b"TTAGGG", b"CCCTAA", b"TTTAGGG", b"AC", b"A", b"ACGTACGTACGTACGTACGTACGTACGTACGT", b"TTAGGGTTAGGGTTAGG",
                  b"ACGTTGCAAGGCTTAGGCATCGATTAGCCATGGACA", b"AC" * 20, b"TTAGGG" * 11 + b"TTA", b"G" * 33):
        seqs = []
        for _ in range(40):
            n = int(rng.integers(0, 40000))
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
            for _k in range(int(rng.integers(0, 6))):
                if n > 200:
                    p = int(rng.integers(0, n - 100))
                    rep = np.frombuffer(motif * int(rng.integers(1, 60)), dtype=np.uint8)
                    rep = rep[: n - p]
                    s[p:p + len(rep)] = rep
            if n > 10 and rng.random() < 0.5:
                s[:: int(rng.integers(2, 50))] |= 0x20   # sprinkle lower case
            seqs.append(s)
        asm = acc.asm_upload(seqs)
        hits = acc.telofind(asm, motif)
        asm.close()
        exp = []
        for ci, s in enumerate(seqs):
            exp += [(ci, int(h["strand"]), int(h["start"]), int(h["end"])) for h in ob.telofind(s, motif)]
        assert [tuple(map(int, h)) for h in hits] == exp, motif


def test_telofind_lookalike_bytes_vs_oracle(acc):
    """copies of the motif in which letters are replaced by other bytes that share bits with them (N and F with G, U and D with
    T, @ and H with A, B and K with C; lower case likewise) must not be reported, copies flush with the contig ends and around
    the tile size (16256) must.  (A 2-bit-code compare with byte-wise confirmation of the candidates was built against this
    test and measured: as many instructions as the automaton — DESIGN.md section 8 — so the automaton stayed.)"""
    rng = np.random.default_rng(4242)
    alias = {ord("A"): b"@HhPp`\x10\x08", ord("C"): b"BKkSs\x02\x03", ord("G"): b"NnFfOo\x06\x07", ord("T"): b"UuDdEe\x04\x05"}
    for motif in (b"TTAGGG", b"CCCTAA", b"TTTAGGG", b"AC", b"G", b"ACGTACGT", b"GGGGGG"):
        seqs = []
        for n in (0, 1, 5, 63, 64, 65, 127, 128, 16255, 16256, 16257, 16256 + 63, 2 * 16256 + 5, 40000, 33000):
            s = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)[rng.integers(0, 10, size=n)].copy()
            for _k in range(12):
                if n > 200:
                    p = int(rng.integers(0, n - 100))
                    rep = np.frombuffer(motif * int(rng.integers(1, 12)), dtype=np.uint8).copy()
                    rep = rep[: n - p]
                    if _k % 3:                                   # look-alikes instead of some letters
                        for i in rng.integers(0, len(rep), size=max(1, len(rep) // 7)):
                            a = alias.get(int(rep[i]) & 0xDF)
                            if a:
                                rep[i] = a[int(rng.integers(0, len(a)))]
                    s[p:p + len(rep)] = rep
            if n >= len(motif):                                  # copies flush with both ends of the contig
                s[:len(motif)] = np.frombuffer(motif, dtype=np.uint8)
                s[n - len(motif):] = np.frombuffer(motif, dtype=np.uint8)
            seqs.append(s)
        asm = acc.asm_upload(seqs)
        hits = acc.telofind(asm, motif)
        asm.close()
        exp = []
        for ci, s in enumerate(seqs):
            exp += [(ci, int(h["strand"]), int(h["start"]), int(h["end"])) for h in ob.telofind(s, motif)]
        assert [tuple(map(int, h)) for h in hits] == exp, motif


# ---------------------------------------------------------------------------------------------------
# coverage windows
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def bg_ctgs(golden_dir):
    return read_bedgraph_pair(os.path.join(golden_dir, "cov-total.bg.gz"), os.path.join(golden_dir, "cov-mq20.bg.gz"))


def gpu_panel_text(acc, ctgs, boring, w=2500, inc=50, L=0.4, H=2.5, Q=0.4, m=1000000, e=100000):
    cov = acc.cov_upload([c[1] for c in ctgs], [c[2] for c in ctgs])
    sd, sq, n = acc.cov_prepare(cov, w, inc)
    assert sd == sum(int(c[1].astype(np.int64).sum()) for c in ctgs)
    assert sq == sum(int(c[2].astype(np.int64).sum()) for c in ctgs)
    mean = int(np.floor(sd / n + 0.5))          # round() of a non-negative double
    lo, hi = acc.cov_threshold(L, mean), acc.cov_threshold(H, mean)
    recs = acc.cov_select(cov, lo, hi, Q, e, m, boring)
    cov.close()
    by = {}
    for r in recs:
        by.setdefault(int(r["ctg"]), []).append(r)
    out = []
    for ci, (name, d, _q) in enumerate(ctgs):
        if not boring:
            if d.size < m:
                out.append(b"%s\t%d\t%d\t.\t.\n" % (name, 0, m))
                continue
            out.append(b"%s\t%d\t%d\t.\t.\n" % (name, 0, e))
            out.append(b"%s\t%d\t%d\t.\t.\n" % (name, d.size - e, d.size))
        for r in by.get(ci, []):
            out.append(b"%s\t%d\t%d\t%d\t%d\n" % (name, r["st"], r["end"], r["depth"], r["mq_depth"]))
    return b"".join(out)


from test_oracle_golden import ABORT_CASES, PANEL_CASES, SPARSE_CASES  # noqa: E402


@pytest.fixture(scope="module")
def sparse_ctgs(golden_dir):
    return read_bedgraph_pair(os.path.join(golden_dir, "sparse-total.bg.gz"), os.path.join(golden_dir, "sparse-mq20.bg.gz"))


@pytest.mark.parametrize("boring,kw,exp", PANEL_CASES)
def test_panel_golden(acc, golden_dir, bg_ctgs, boring, kw, exp):
    assert gpu_panel_text(acc, bg_ctgs, boring, **kw) == golden(golden_dir, exp)


@pytest.mark.parametrize("boring,kw,exp", SPARSE_CASES)
def test_panel_golden_sparse_windows(acc, golden_dir, sparse_ctgs, boring, kw, exp):
    """-i larger than -w (src/boringbits_main.c:338-369: disjoint windows, each the head of its own block on the device), -e beyond every
    contig, w % inc = 1 with a contig of exactly w + 51: reference stdout"""
    assert gpu_panel_text(acc, sparse_ctgs, boring, **kw) == golden(golden_dir, exp)


@pytest.mark.parametrize("pair,boring,kw", ABORT_CASES)
def test_cov_prepare_reports_the_asserts_of_get_regs(acc, bg_ctgs, sparse_ctgs, pair, boring, kw):
    """where the reference dies of assert(st<end) (:353) the C ABI says so (CORNETTO_E_ASSERT = -7) and names the line; the CLI turns
    that into the reference's SIGABRT (tests/test_gpu_cli.py)"""
    import cornetto_amd
    ctgs = bg_ctgs if pair == "cov" else sparse_ctgs
    cov = acc.cov_upload([c[1] for c in ctgs], [c[2] for c in ctgs])
    with pytest.raises(cornetto_amd.AccelError) as ei:
        acc.cov_prepare(cov, kw["w"], kw["inc"])
    assert ei.value.status == -7 and "boringbits_main.c:353" in str(ei.value)
    sd, _sq, _n = acc.cov_prepare(cov, 2500, 50)      # the object is still good
    assert sd == sum(int(c[1].astype(np.int64).sum()) for c in ctgs)
    cov.close()


@pytest.mark.parametrize("w,inc", [(64, 1000), (300, 301), (1, 2), (50, 127), (50, 128), (50, 129), (2500, 4000), (7, 50)])
def test_cov_regs_sparse_windows_vs_oracle(acc, w, inc):
    """inc > w on lengths the reference gets through (every length <= w, or in (k inc, k inc + w]): both block-sum kernels (LDS staging up
    to inc = 128, direct loads beyond), heads clipped by the contig's end"""
    rng = np.random.default_rng(w * 977 + inc)
    lens = sorted({1, w, max(1, w - 1), inc + 1, inc + w, 2 * inc + 1, 5 * inc + max(1, w // 2), 300 * inc + 1, 300 * inc + w, 1000 * inc + 1 + w // 3})
    assert all(ob.regs_assert(n, w, inc) == 0 for n in lens)
    depths = [rng.integers(0, 65536, size=n).astype(np.uint16) for n in lens]
    mqs = [rng.integers(0, 65536, size=n).astype(np.uint16) for n in lens]
    cov = acc.cov_upload(depths, mqs)
    sd, sq, n = acc.cov_prepare(cov, w, inc)
    assert (sd, sq, n) == (sum(int(d.astype(np.int64).sum()) for d in depths), sum(int(q.astype(np.int64).sum()) for q in mqs), sum(lens))
    for ci in range(len(lens)):
        got = acc.cov_regs(cov, ci)
        exp = ob.get_regs(depths[ci], mqs[ci], w, inc)
        assert np.array_equal(got, exp.astype(got.dtype)), (w, inc, lens[ci])
    cov.close()


@pytest.mark.parametrize("w,inc", [(2500, 50), (300, 7), (1000, 1000), (2500, 49), (64, 1), (5000, 130), (777, 200), (50, 50)])
def test_cov_regs_vs_oracle(acc, w, inc):
    rng = np.random.default_rng(w * 131 + inc)
    lens = [1, 2, 49, 50, 51, 120, 2449, 2450, 2451, 2500, 2501, 2551, 12800, 12801, 30000, 77777]
    depths = [rng.integers(0, 65536, size=n).astype(np.uint16) for n in lens]
    mqs = [rng.integers(0, 65536, size=n).astype(np.uint16) for n in lens]
    cov = acc.cov_upload(depths, mqs)
    sd, sq, n = acc.cov_prepare(cov, w, inc)
    assert (sd, sq, n) == (sum(int(d.astype(np.int64).sum()) for d in depths), sum(int(q.astype(np.int64).sum()) for q in mqs), sum(lens))
    for ci in range(len(lens)):
        got = acc.cov_regs(cov, ci)
        exp = ob.get_regs(depths[ci], mqs[ci], w, inc)
        assert np.array_equal(got, exp.astype(got.dtype)), (w, inc, lens[ci])
    cov.close()


@pytest.mark.parametrize("w,inc", [(2500, 50), (300, 7)])
def test_cov_select_more_windows_than_the_first_block_holds(w, inc):
    """a fresh handle sizes the ordered block for an eighth of the windows (at least 65 536): a selection that takes nearly all of 120 000+
    is detected after the fact and rerun at the exact size — both record forms, then again on the grown block; against the oracle's windows
    and the reference's predicate (src/boringbits_main.c:467,473-474)"""
    rng = np.random.default_rng(w + inc)
    lens = [int(inc * 125_000 + 17), 30_000, 12_345]
    depths = [rng.integers(20, 41, size=n).astype(np.uint16) for n in lens]
    mqs = [np.minimum(d, rng.integers(15, 41, size=d.size)).astype(np.uint16) for d in depths]
    import cornetto_amd
    a = cornetto_amd.Accel(0)
    cov = a.cov_upload(depths, mqs)
    a.cov_prepare(cov, w, inc)
    lo, hi, Q, edge, min_len = 12, 75, 0.4, 1000, 10_000
    exp = []
    for ci, (d, q) in enumerate(zip(depths, mqs)):
        regs = ob.get_regs(d, q, w, inc)
        for r in regs:
            st, end, dep, mq = int(r["st"]), int(r["end"]), int(r["depth"]), int(r["mq_depth"])
            if len(d) > min_len and st > edge and end < len(d) - edge and not ob.is_fun(dep, mq, lo, hi, Q):
                exp.append((ci, st, end, dep, mq))
    assert len(exp) > 100_000
    for _ in range(2):
        recs = a.cov_select(cov, lo, hi, Q, edge, min_len, True)
        assert [tuple(int(x) for x in r) for r in recs] == exp
        pk, cf = a.cov_select_packed(cov, lo, hi, Q, edge, min_len, True)
        un = a.unpack_regs(pk, cf, lens, w)
        assert [tuple(int(x) for x in r) for r in un] == exp
    cov.close()
    a.close()


@pytest.mark.parametrize("w,inc", [(40_000, 1000), (70_000, 50), (33_000, 33_001)])
def test_cov_select_windows_whose_int_sums_wrap(w, inc):
    """windows beyond 32768 positions of depth up to 65535: the reference's `int` sums wrap (gcc -O2: two's complement; the oracle restates that) and
    a mean can be negative — it does not fit the packed 16-bit form, so the raw records stay full-size on the device, the packed interface refuses,
    and the unpacked selection equals the oracle's windows under the reference's predicate (src/boringbits_main.c:439,467,473-474)"""
    import cornetto_amd
    rng = np.random.default_rng(w + inc)
    lens = [3 * w + 17, w + 1, 2 * inc + (w if inc > w else 5), 1200]
    lens = [n for n in lens if ob.regs_assert(n, w, inc) == 0]
    depths = [rng.integers(60_000, 65_536, size=n).astype(np.uint16) for n in lens]
    mqs = [np.minimum(d, rng.integers(30_000, 65_536, size=d.size)).astype(np.uint16) for d in depths]
    a = cornetto_amd.Accel(0)
    cov = a.cov_upload(depths, mqs)
    a.cov_prepare(cov, w, inc)
    lo, hi, Q, edge, min_len = 100, 64_000, 0.6, 10, 1000
    assert any(int(r["depth"]) < 60_000 for r in ob.get_regs(depths[0], mqs[0], w, inc)) or w < 36_000      # (the wrapped sums: means far from the 60 000-65 535 of the data)
    for boring in (False, True):
        exp = []
        for ci, (d, q) in enumerate(zip(depths, mqs)):
            n = d.size
            if (n > min_len) if boring else (n >= min_len):
                for r in ob.get_regs(d, q, w, inc):
                    st, end, dep, mq = int(r["st"]), int(r["end"]), int(r["depth"]), int(r["mq_depth"])
                    fun = bool(ob.is_fun(dep, mq, lo, hi, Q))
                    if (st > edge and end < n - edge and not fun) if boring else fun:
                        exp.append((ci, st, end, dep, mq))
        recs = a.cov_select(cov, lo, hi, Q, edge, min_len, boring)
        assert [tuple(int(x) for x in r) for r in recs] == exp, boring
    with pytest.raises(cornetto_amd.AccelError) as ei:
        a.cov_select_packed(cov, lo, hi, Q, edge, min_len, False)
    assert ei.value.status == -5
    cov.close()
    a.close()


def test_sdust_repeatable_with_many_small_chunks(dacc, golden_dir, monkeypatch):
    """the chunk queue hands the chunks to different lanes / waves at different times on every run: 24 runs over
    ~100 k tiny chunks must all give the golden answer (a build of the kernel that spilled registers did not)"""
    acc = dacc                                   # the development build: the switches below exist there only
    recs = _records(golden_dir, "mix.fa.gz")
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", "16")
    exp = golden(golden_dir, "mix.sdust.exp")
    for it in range(24):
        assert gpu_sdust_text(acc, recs, 20, 64) == exp, it


def test_sdust_queue_and_order_do_not_change_results(dacc, golden_dir, monkeypatch):
    """the chunk queue (persistent waves), its low-complexity-first order and the grid size are scheduling only"""
    acc = dacc                                   # the development build: the switches below exist there only
    recs = _records(golden_dir, "mix.fa.gz")
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", "100")
    exp = golden(golden_dir, "mix.sdust.exp")
    for waves, order, runon in (("1", "1", "1"), ("2", "0", "1"), ("5", "1", "0"), ("100000", "0", "0"), ("3", "1", "1")):
        monkeypatch.setenv("CORNETTO_SDUST_WAVES", waves)
        monkeypatch.setenv("CORNETTO_SDUST_ORDER", order)
        monkeypatch.setenv("CORNETTO_SDUST_RUNON", runon)      # lanes run on into the next chunk while it is free
        for chunk in ("100", "37", "64"):
            monkeypatch.setenv("CORNETTO_SDUST_CHUNK", chunk)
            assert gpu_sdust_text(acc, recs, 20, 64) == exp, (waves, order, runon, chunk)
    # a table of two chunk sizes: the last 35 % of the bases in chunks 2 / 4 times shorter, handed out last
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", "512")
    for tail, div, dense in (("35", "4", "2"), ("35", "2", "0"), ("90", "3", "2"), ("5", "16", "1")):
        monkeypatch.setenv("CORNETTO_SDUST_TAIL", tail)
        monkeypatch.setenv("CORNETTO_SDUST_TAILDIV", div)
        monkeypatch.setenv("CORNETTO_SDUST_DENSE", dense)
        assert gpu_sdust_text(acc, recs, 20, 64) == exp, (tail, div, dense)


@pytest.mark.parametrize("chunk", ["64", "500", "1536", "1792"])
def test_sdust_long_word_free_stretches_vs_oracle(dacc, monkeypatch, chunk):
    """more than 1024 bases without W-2 word emissions before a chunk: the lane's local backward scan gives
    up, the host builds the word-count table and reruns (sdust.hip, SD_SCAN_CAP) — still exact, including the
    stale-window quirk across the N-dense stretch"""
    acc = dacc                                   # the development build: the switches below exist there only
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", chunk)
    rng = np.random.default_rng(int(chunk))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rnd(n):
        return acgt[rng.integers(0, 4, size=n)].tobytes()

    seqs = [
        rnd(3000) + b"N" * 5000 + rnd(3000),
        b"AC" * 40 + b"ACN" * 1500 + b"AC" * 60 + rnd(500) + b"N" * 2000 + b"ACGN" * 800 + b"A" * 100 + rnd(2000),
        b"N" * 4000 + b"ACACACACACACACACACACACAC" + b"N" * 3000 + b"ACACACACACACAC" + rnd(1000),
        (b"ACG" + b"N" * 40) * 100 + b"T" * 80 + rnd(800),           # one word every 43 bases: 62 words span 2.6 kb
        rnd(200) + b"ACN" * 3000,
        b"ACN" * 3000,
    ]
    seqs = [np.frombuffer(s, dtype=np.uint8) for s in seqs]
    asm = acc.asm_upload(seqs)
    iv = acc.sdust(asm, 20, 64)
    iv2 = acc.sdust(asm, 20, 64)            # second call reuses the cached table
    asm.close()
    got = [(int(x["ctg"]), int(x["start"]), int(x["finish"])) for x in iv]
    assert got == [(int(x["ctg"]), int(x["start"]), int(x["finish"])) for x in iv2]
    exp = []
    for ci, s in enumerate(seqs):
        for r in ob.sdust(s, 20, 64):
            r = int(r)
            exp.append((ci, r >> 32, r & 0xFFFFFFFF))
    assert got == exp


# ---------------------------------------------------------------------------------------------------
# bedgraph ingest on the device
# ---------------------------------------------------------------------------------------------------
def _bg_text(golden_dir):
    import gzip
    t = gzip.open(os.path.join(golden_dir, "cov-total.bg.gz")).read()
    q = gzip.open(os.path.join(golden_dir, "cov-mq20.bg.gz")).read()
    return t, q


def _split(data, rng, mean):
    cuts = sorted(set(int(x) for x in rng.integers(0, len(data) + 1, size=max(1, len(data) // mean))))
    out, prev = [], 0
    for c in cuts + [len(data)]:
        out.append(data[prev:c])
        prev = c
    return out


@pytest.mark.parametrize("mean", [10**9, 200000, 3000, 37])
def test_bedgraph_ingest_matches_host_parse(acc, golden_dir, bg_ctgs, mean):
    """any split of the two byte streams gives the arrays the reference's reader builds"""
    t, q = _bg_text(golden_dir)
    if mean < 1000:                       # tiny pieces: keep the run short
        nlines = 4000
        t = b"".join(t.splitlines(True)[:nlines])
        q = b"".join(q.splitlines(True)[:nlines])
    rng = np.random.default_rng(mean)
    cov, names, ncl = acc.bedgraph_ingest(_split(t, rng, mean), _split(q, rng, mean + 7))
    exp = bg_ctgs
    if mean < 1000:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            open(os.path.join(d, "t.bg"), "wb").write(t)
            open(os.path.join(d, "q.bg"), "wb").write(q)
            exp = read_bedgraph_pair(os.path.join(d, "t.bg"), os.path.join(d, "q.bg"))
    assert names == [c[0] for c in exp]
    assert list(cov.lens) == [len(c[1]) for c in exp]
    w, inc = 64, 1                          # window = the position itself would need w=1; use sums to compare arrays
    sd, sq, n = acc.cov_prepare(cov, w, inc)
    assert sd == sum(int(c[1].astype(np.int64).sum()) for c in exp)
    assert sq == sum(int(c[2].astype(np.int64).sum()) for c in exp)
    assert n == sum(len(c[1]) for c in exp)
    for ci, c in enumerate(exp):            # every window of every contig == oracle on the host-parsed arrays
        got = acc.cov_regs(cov, ci)
        ex = ob.get_regs(c[1], c[2], w, inc)
        assert np.array_equal(got, ex.astype(got.dtype)), ci
    assert ncl == sum(int((np.array(v) > 65535).sum()) for v in _raw_depths(t, q))
    cov.close()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("mean", [200000, 3000])
def test_bedgraph_ingest_with_the_next_pieces_on_their_way(acc, golden_dir, mean, mode):
    """cornetto_bgin_prefetch(): pieces staged on the device ahead of their feed (mode 1, as the CLI does) and prefetches that no feed takes (mode 2)
    give the coverage the feeds alone give"""
    t, q = _bg_text(golden_dir)
    rng = np.random.default_rng(mean)
    tp, qp = _split(t, rng, mean), _split(q, rng, mean + 7)
    a, na, ca = acc.bedgraph_ingest(tp, qp)
    b, nb, cb = acc.bedgraph_ingest(tp, qp, prefetch=mode)
    try:
        assert na == nb and ca == cb and list(a.lens) == list(b.lens)
        assert acc.cov_prepare(a, 300, 7) == acc.cov_prepare(b, 300, 7)
        for ci in range(len(a.lens)):
            assert np.array_equal(acc.cov_regs(a, ci), acc.cov_regs(b, ci)), ci
    finally:
        a.close()
        b.close()


def test_accel_warm_and_text_api_arguments(acc):
    """cornetto_accel_warm() leaves the handle as it found it; cornetto_text_* refuses what it cannot take"""
    import ctypes as C
    import cornetto_amd
    L = acc.L
    rng = np.random.default_rng(9)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 50000)].copy()
    seq[1000:1600] = np.frombuffer(b"TTAGGG" * 100, dtype=np.uint8)
    asm = acc.asm_upload([seq])
    try:
        before = (acc.sdust(asm, 20, 64).copy(), acc.telofind(asm, b"TTAGGG").copy())
        acc.warm(7)
        acc.warm(1)
        after = (acc.sdust(asm, 20, 64), acc.telofind(asm, b"TTAGGG"))
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    finally:
        asm.close()
    t = C.c_void_p()
    assert L.cornetto_text_open(acc.h, 0, C.byref(t)) == -3 and L.cornetto_text_open(acc.h, 1 << 33, C.byref(t)) == -3        # CORNETTO_E_ARG
    assert L.cornetto_text_open(acc.h, 1000, C.byref(t)) == 0
    slab = L.cornetto_pinned_alloc(4096)
    try:
        assert L.cornetto_text_put(acc.h, t, slab, 600, 500, 0) == -3             # beyond the capacity
        assert L.cornetto_text_put(acc.h, t, slab, 10, 0, 4) == -3                # no such queue
        C.memmove(slab, b">a\nACGT\n>b\nGG\n", 14)
        assert L.cornetto_text_put(acc.h, t, slab, 14, 0, 3) == 0 and L.cornetto_text_wait(acc.h, t, 3) == 0
        p, cnt, used, plain = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int32()
        assert L.cornetto_fasta_split_text(acc.h, t, 2000, 1, C.byref(p), C.byref(cnt), C.byref(used), C.byref(plain), None) == -3
        assert L.cornetto_fasta_split_text(acc.h, t, 14, 1, C.byref(p), C.byref(cnt), C.byref(used), C.byref(plain), None) == 0
        assert (cnt.value, used.value, plain.value) == (2, 14, 1)
        L.cornetto_free(p)
    finally:
        L.cornetto_text_free(acc.h, t)
        L.cornetto_pinned_free(slab)


def _raw_depths(t, q):
    return ([int(l.split()[3]) for l in t.splitlines()], [int(l.split()[3]) for l in q.splitlines()])


def test_bedgraph_ingest_errors(acc, golden_dir):
    import cornetto_amd
    t, q = _bg_text(golden_dir)
    tl, ql = t.splitlines(True)[:200], q.splitlines(True)[:200]

    def kind(tt, qq, pieces=1):
        tt, qq = b"".join(tt), b"".join(qq)
        tp = [tt[i::1] for i in [0]] if pieces == 1 else _split(tt, np.random.default_rng(1), 50)
        qp = [qq] if pieces == 1 else _split(qq, np.random.default_rng(2), 50)
        try:
            cov, _n, _c = acc.bedgraph_ingest(tp, qp)
            cov.close()
            return 0, -1
        except cornetto_amd.BedgraphFormatError as e:
            return e.kind, e.record

    for pieces in (1, 9):
        assert kind(tl, ql, pieces) == (0, -1)
        assert kind([b"track type=bedGraph\n"] + tl, [b"track type=bedGraph\n"] + ql, pieces)[0] == 1      # 4 columns
        assert kind(tl, ql[:150], pieces) == (3, 150)                                                        # cov-mq ends first
        assert kind(tl[:150], ql, pieces) == (0, -1)                                                         # extra cov-mq records are ignored
        assert kind(tl, ql[:40] + ql[41:], pieces)[0] == 3                                                   # not in the same order
        assert kind(tl[:20] + tl[21:], ql[:20] + ql[21:], pieces) == (4, 20)                                 # not incremental
        assert kind([b"ptg000001l\t0\t5\t30\n"], [b"ptg000001l\t0\t5\t30\n"], pieces) == (5, 0)              # end != start + 1
        assert kind(tl + [b"ptg000001l\t7\n"], ql + [b"ptg000001l\t7\t8\t1\n"], pieces) == (1, 200)          # trailing partial record
        assert kind(tl[:50] + [b"x\t1\t2\tabc\n"], ql[:51], pieces) == (1, 50)
        assert kind(tl[:51], ql[:50] + [b"ptg000001l\t50\t51\n"], pieces) == (2, 50)
    # records need not be lines: four white-space separated tokens, as fscanf reads them
    flat_t = b" ".join(b"".join(tl).split()) + b"\n"
    flat_q = b"\n".join(b"".join(ql).split())
    assert kind([flat_t], [flat_q]) == (0, -1)
    assert kind([b""], [b""]) == (0, -1)
    assert kind([b"\n\n  \n"], [b""]) == (0, -1)


def test_timing_levels(acc):
    """cornetto_accel_set_timing: which launches carry HIP event pairs (results never depend on it)"""
    rng = np.random.default_rng(3)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 200000)]
    asm = acc.asm_upload([seq])
    try:
        acc.set_timing(2)
        ref = acc.sdust(asm, 20, 64)
        names2 = {n for n, _ in acc.last_timing()}
        acc.set_timing(1)
        one = acc.sdust(asm, 20, 64)
        names1 = {n for n, _ in acc.last_timing()}
        acc.set_timing(0)
        zero = acc.sdust(asm, 20, 64)
        names0 = {n for n, _ in acc.last_timing()}
    finally:
        acc.set_timing(2)
        asm.close()
    assert np.array_equal(ref, one) and np.array_equal(ref, zero)
    assert names0 == set() and names1 == {"sdust_kernel"} and "sdust_scan" in names2 and "sdust_kernel" in names2
    with pytest.raises(Exception):
        acc.set_timing(3)


def test_lazy_result_copies_equal_the_synchronous_ones(acc):
    """cornetto_accel_set_lazy / cornetto_accel_wait: the telomere runs and the selected windows are copied on a stream of their own while the
    next calls of the handle run; after the wait the arrays hold what the synchronous calls return"""
    rng = np.random.default_rng(77)
    lens = [1_200_000, 300_000, 2_000_000, 70_000, 1_000_001]
    seqs = []
    for n in lens:
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
        for _ in range(40):                                              # telomere runs, both strands
            p = int(rng.integers(0, n - 700))
            unit = np.frombuffer(b"TTAGGG" if rng.integers(0, 2) else b"CCCTAA", dtype=np.uint8)
            s[p:p + 600] = np.tile(unit, 100)
        seqs.append(s)
    depths = [rng.integers(0, 80, size=n).astype(np.uint16) for n in lens]
    mqs = [np.minimum(depths[i], rng.integers(0, 80, size=n)).astype(np.uint16) for i, n in enumerate(lens)]
    asm, cov = acc.asm_upload(seqs), acc.cov_upload(depths, mqs)
    thr = acc.telowin_threshold(0.4, 99.9)

    def run():
        hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
        acc.cov_prepare(cov, 2500, 50)
        pk, first = acc.cov_select_packed(cov, 39, 40, 0.4, 100000, 1000000, False)     # (window means are 39.5 +- 0.5: about half of them)
        return hits, wins, pk, first
    try:
        ref = [x.copy() for x in run()]
        acc.set_lazy(True)
        for _ in range(3):
            got = run()
            acc.wait()
            for g, r in zip(got, ref):
                assert np.array_equal(g, r)
        acc.wait()                                                       # nothing pending: returns at once
        got = run()
        acc.set_lazy(False)                                              # turning it off waits as well
        for g, r in zip(got, ref):
            assert np.array_equal(g, r)
    finally:
        acc.set_lazy(False)
        asm.close()
        cov.close()
    assert len(ref[0]) > 100 and len(ref[2]) > 1000


@pytest.mark.parametrize("sift", ["1", "0"])
def test_launch_count_moves_with_every_resident_sdust_launch(dacc, monkeypatch, sift):
    """cornetto_accel_launch_count: readable from another thread while the call runs; bench.py waits for it to move before the other
    stream's first kernel (where the resident waves land decides how much room that stream finds on every CU)"""
    acc = dacc                                   # the development build: the switches below exist there only
    import threading
    monkeypatch.setenv("CORNETTO_SDUST_SIFT", sift)
    rng = np.random.default_rng(11)
    asm = acc.asm_upload([_sift_stress_seq(rng, 8_000_000, 1)])
    try:
        c0 = acc.launch_count()
        seen = []
        done = threading.Event()

        def watch():
            while not done.is_set():
                seen.append(acc.launch_count())
        t = threading.Thread(target=watch)
        t.start()
        a = acc.sdust(asm, 20, 64)
        b = acc.sdust(asm, 20, 64)
        done.set()
        t.join()
        assert acc.launch_count() == c0 + 2 and np.array_equal(a, b)
        assert all(c0 <= x <= c0 + 2 for x in seen) and seen == sorted(seen)
    finally:
        asm.close()
