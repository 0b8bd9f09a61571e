"""Device FASTQ record splitter (cornetto_fastq_split) against the oracle's restatement of
kseq's record framing (oracle.c orc_fastx_parse, itself pinned by the reference's seq / fa2bed outputs)."""
import os

import numpy as np
import pytest

import oracle_bind as ob
from helpers import tricky_fastx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def acc():
    import cornetto_amd
    a = cornetto_amd.Accel(0)
    yield a
    a.close()


def fields(text, recs):
    """(name, comment, seq, qual) byte strings of the device's records"""
    out = []
    for r in recs:
        h, nl, cl, L = int(r["head"]), int(r["name_len"]), int(r["comment_len"]), int(r["len"])
        name = text[h + 1:h + 1 + nl]
        com = text[h + nl + 2:h + nl + 2 + cl] if cl else b""
        out.append((name, com, text[int(r["seq"]):int(r["seq"]) + L], text[int(r["qual"]):int(r["qual"]) + L]))
    return out


def check_prefix(acc, text, final=True):
    """the device's records must be the leading records of kseq's reading, and `plain` must tell whether kseq's next
    record starts exactly where the device stopped being sure"""
    recs, used, plain, _ = acc.fastq_split(text, final=final)
    exp, rc = ob.fastx_parse(text)
    got = fields(text, recs)
    assert got == [(n, c, s, q) for n, c, s, q in exp[:len(got)]], (text[:200], got[:3], exp[:3])
    assert all(q is not None for _, _, _, q in exp[:len(got)])
    assert 0 <= used <= len(text)
    if len(got):
        assert used == int(recs[-1]["qual"]) + len(text[int(recs[-1]["qual"]):].split(b"\n", 1)[0]) + (1 if b"\n" in text[int(recs[-1]["qual"]):] else 0)
    # the rest, read sequentially, continues the same record list
    rest, rc2 = ob.fastx_parse(text[used:])
    assert [(n, c, s, q) for n, c, s, q in exp[len(got):]] == rest and rc2 == rc
    return recs, used, plain, exp


def test_reads_golden_all_plain(acc, golden_dir):
    text = open(os.path.join(golden_dir, "reads.fq"), "rb").read()
    recs, used, plain, exp = check_prefix(acc, text)
    assert len(recs) == len(exp) == 40 and used == len(text) and plain


def test_strict_text_is_indexed_completely(acc):
    rng = np.random.default_rng(11)
    for it in range(40):
        text = tricky_fastx(rng, int(rng.integers(1, 60)), strict=True)
        if not text.endswith(b"\n") and text.endswith(b"+"):
            text += b"\n"
        recs, used, plain, exp = check_prefix(acc, text)
        if text.endswith(b"+\n") or text.endswith(b"+\r\n"):    # empty last read whose (empty) quality line was cut off with the newline
            continue
        assert len(recs) == len(exp) and used == len(text) and plain, it


def test_irregular_text_stops_at_the_first_irregular_record(acc):
    rng = np.random.default_rng(12)
    n_stop = 0
    for it in range(120):
        text = tricky_fastx(rng, int(rng.integers(0, 25)), strict=False)
        recs, used, plain, exp = check_prefix(acc, text)
        if len(recs) < len(exp):
            assert not plain
            n_stop += 1
    assert n_stop > 30


def test_pieces_any_split_point(acc):
    """feed a text in pieces the way a file reader does: consumed bytes are dropped, the rest is handed over again"""
    rng = np.random.default_rng(13)
    text = tricky_fastx(rng, 300, strict=True)
    if not text.endswith(b"\n"):
        text += b"\n"
    exp, rc = ob.fastx_parse(text)
    for piece in (1 << 20, 4096, 1500, 977):
        got, pos, pend = [], 0, b""
        while True:
            chunk = text[pos:pos + piece]
            pos += len(chunk)
            final = pos >= len(text)
            buf = pend + chunk
            recs, used, plain, _ = acc.fastq_split(buf, final=final)
            assert plain
            got += fields(buf, recs)
            pend = buf[used:]
            if final:
                break
        assert pend == b"" and got == [(n, c, s, q) for n, c, s, q in exp], piece


def test_empty_and_tiny_inputs(acc):
    for text in (b"", b"\n", b"@", b"@r\n", b"@r\nA\n+\n", b"@r\nA\n+\nI", b"@r\nA\n+\nI\n", b"@r\n\n+\n\n", b"\n\n\n\n"):
        check_prefix(acc, text)
    recs, used, plain, _ = acc.fastq_split(b"@r\nA\n+\n", final=False)
    assert len(recs) == 0 and used == 0 and plain


def test_reads_resident_for_sdust(acc, golden_dir):
    """`seq -m` + per-read sdust without a host-side copy of the reads: same intervals as uploading the kept reads"""
    text = open(os.path.join(golden_dir, "reads.fq"), "rb").read()
    for min_len in (0, 100, 5000):
        recs, used, plain, reads = acc.fastq_split(text, min_len=min_len, want_reads=True)
        exp, _ = ob.fastx_parse(text)
        kept = [s for _, _, s, _ in exp if len(s) >= min_len]
        assert list(recs["keep"]) == [1 if len(s) >= min_len else 0 for _, _, s, _ in exp]
        assert [int(x) for x in reads.lens] == [len(s) for s in kept]
        got = acc.sdust(reads, 20, 64)
        want = []
        for i, s in enumerate(kept):
            for v in ob.sdust(np.frombuffer(s, dtype=np.uint8), 20, 64):
                want.append((i, int(v) >> 32, int(v) & 0xFFFFFFFF))
        assert [(int(a), int(b), int(c)) for a, b, c in got] == want, min_len
        reads.close()


def test_large_piece_many_reads(acc):
    """2 M short reads + long reads in one piece: record table against a numpy restatement of the four-line rule"""
    rng = np.random.default_rng(14)
    n = 200000
    lens = np.concatenate([rng.integers(1, 300, n), rng.integers(5000, 20000, 200)])
    rng.shuffle(lens)
    parts = []
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i, L in enumerate(lens.tolist()):
        parts.append(b"@r%d ch=%d\n" % (i, i % 512))
        s = alpha[rng.integers(0, 4, L)].tobytes()
        parts.append(s + b"\n+\n" + b"I" * L + b"\n")
    text = b"".join(parts)
    recs, used, plain, reads = acc.fastq_split(text, min_len=250, want_reads=True)
    assert plain and used == len(text) and len(recs) == len(lens)
    assert np.array_equal(recs["len"], lens.astype(np.int32))
    assert np.array_equal(recs["keep"], (lens >= 250).astype(np.int32))
    idx = rng.integers(0, len(lens), 200)
    f = fields(text, recs[idx])
    for k, i in enumerate(idx.tolist()):
        assert f[k][0] == b"r%d" % i and f[k][1] == b"ch=%d" % (i % 512) and len(f[k][2]) == lens[i] and f[k][3] == b"I" * int(lens[i])
    # the packed reads are the kept reads: sdust of a few of them equals sdust of the bytes
    got = acc.sdust(reads, 20, 64)
    kept = np.flatnonzero(lens >= 250)
    for j in rng.integers(0, len(kept), 20).tolist():
        i = int(kept[j])
        s = np.frombuffer(text[int(recs[i]["seq"]):int(recs[i]["seq"]) + int(lens[i])], dtype=np.uint8)
        want = [(int(v) >> 32, int(v) & 0xFFFFFFFF) for v in ob.sdust(s, 20, 64)]
        assert [(int(b), int(c)) for a, b, c in got[got["ctg"] == j]] == want
    reads.close()


# ---- through the CLI: `cornetto sdust reads.fastq` frames the records on the device ---------------------------------
def test_config5_generator_100_mbases_vs_oracle(acc):
    """BASELINE config 5 at 100 Mbases: the SURVEY 8d generator (synth.make_fastq_piece: log-normal lengths 200 .. 200 000, every
    fourth read with a homopolymer stretch) -> cornetto_fastq_split with the length test of `seq -m 10000` (src/seq.c:120) ->
    sdust per kept read; the oracle frames the same text (kseq restatement), filters by length and runs sdust read by read"""
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cornetto_amd import synth
    from concurrent.futures import ThreadPoolExecutor
    text_d, lens, off = synth.make_fastq_piece(torch, torch.device("cuda", 0), 100e6, 77)
    text = text_d.cpu().numpy()
    del text_d
    assert lens.min() >= 200 and lens.max() <= 200_000 and 95e6 < lens.sum() < 105e6
    recs, used, plain, reads = acc.fastq_split(text, final=True, min_len=10000, want_reads=True)
    assert plain and used == text.size and len(recs) == len(lens) and np.array_equal(recs["len"], lens.astype(np.int32))
    assert np.array_equal(recs["keep"] == 1, lens >= 10000)
    iv = acc.sdust(reads, 20, 64)
    reads.close()
    kept = np.nonzero(lens >= 10000)[0]
    hl = len(synth.FQ_HEAD % (0, 0))
    ob.lib()

    def one(i):
        a = int(off[i]) + hl
        return np.asarray(ob.sdust(text[a:a + int(lens[i])], 20, 64), dtype=np.uint64)

    with ThreadPoolExecutor(max(1, min(16, (os.cpu_count() or 2) - 1))) as ex:
        exp = list(ex.map(one, kept.tolist()))
    got_c = iv["ctg"].astype(np.int64)
    got = (iv["start"].astype(np.uint64) << np.uint64(32)) | iv["finish"].astype(np.uint32).astype(np.uint64)
    assert len(got) == sum(len(e) for e in exp) and len(got) > len(kept) // 8
    assert np.array_equal(got, np.concatenate(exp))
    assert np.array_equal(got_c, np.concatenate([np.full(len(e), k, dtype=np.int64) for k, e in enumerate(exp)]))
    # the framing itself against the oracle's kseq restatement, on the first 5 MB
    cut = int(off[np.searchsorted(off, 5_000_000)])
    ex_recs, rc = ob.fastx_parse(text[:cut])
    assert rc == -1 and [len(r[2]) for r in ex_recs] == lens[:len(ex_recs)].tolist()


def run_cli(args, env=None, data=None):
    import subprocess
    import cornetto_amd
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([cornetto_amd.CLI_PATH] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    return p.returncode, p.stdout, p.stderr


def sdust_text(text):
    """stdout of the reference's `sdust` for this FASTA/FASTQ text: kseq records, intervals per record (sdust.c:196-203)"""
    recs, rc = ob.fastx_parse(text)
    out = []
    for name, _, seq, _ in recs:
        for v in ob.sdust(np.frombuffer(seq, dtype=np.uint8), 20, 64):
            out.append(b"%s\t%d\t%d\n" % (name, int(v) >> 32, int(v) & 0xFFFFFFFF))
    return b"".join(out)


@pytest.mark.parametrize("piece", [None, "64", "700", "5000", "100000"])
def test_cli_sdust_reads_golden_any_piece_size(golden_dir, piece):
    env = {"CORNETTO_FASTQ_PIECE": piece} if piece else None
    rc, out, err = run_cli(["sdust", os.path.join(golden_dir, "reads.fq")], env=env)
    assert rc == 0, err.decode()
    assert out == open(os.path.join(golden_dir, "reads.sdust.exp"), "rb").read()


def lowcomplex_fastx(rng, n_rec, strict):
    return tricky_fastx(rng, n_rec, strict=strict, lowc=0.6)


def test_cli_sdust_irregular_fastq_falls_back_mid_file(tmp_path):
    rng = np.random.default_rng(21)
    f = str(tmp_path / "t.fq")
    n_out = 0
    for it in range(12):
        # a plain head (so that the device path is taken), then irregular records
        text = lowcomplex_fastx(rng, 30, True)
        if not text.endswith(b"\n"):
            text += b"\n"
        text += lowcomplex_fastx(rng, 30, False)
        open(f, "wb").write(text)
        want = sdust_text(text)
        n_out += len(want)
        for piece in ("1500", "4096", None):
            rc, out, err = run_cli(["sdust", f], env={"CORNETTO_FASTQ_PIECE": piece} if piece else None)
            assert rc == 0 and out == want, (it, piece, err.decode()[-300:])
    assert n_out > 1000


def test_cli_sdust_fastq_gz_and_stdin(tmp_path, golden_dir):
    import gzip
    text = open(os.path.join(golden_dir, "reads.fq"), "rb").read()
    want = open(os.path.join(golden_dir, "reads.sdust.exp"), "rb").read()
    gz = str(tmp_path / "reads.fq.gz")
    with gzip.open(gz, "wb") as fh:
        fh.write(text)
    rc, out, err = run_cli(["sdust", gz], env={"CORNETTO_FASTQ_PIECE": "3000"})
    assert rc == 0 and out == want
    rc, out, err = run_cli(["sdust", "-"], data=text, env={"CORNETTO_FASTQ_PIECE": "3000"})
    assert rc == 0 and out == want


def test_cli_sdust_record_larger_than_a_piece_and_truncated_quality(tmp_path):
    rng = np.random.default_rng(22)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    big = alpha[rng.integers(0, 2, 9000)].tobytes()
    text = b"@a\nACGTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTACGT\n+\n" + b"I" * 45 + b"\n@big x\n" + big + b"\n+\n" + b"#" * 9000 + b"\n" \
           b"@c\nAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n+\n" + b"I" * 45 + b"\n"
    f = str(tmp_path / "big.fq")
    open(f, "wb").write(text)
    for piece in ("256", "4000", None):
        rc, out, err = run_cli(["sdust", f], env={"CORNETTO_FASTQ_PIECE": piece} if piece else None)
        assert rc == 0 and out == sdust_text(text), piece
    # kseq_read returns -2 at a quality string of another length and the reference's loop ends there (sdust.c:196)
    cut = b"@a\nAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n+\n" + b"I" * 45 + b"\n@b\nAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n+\nII\n" \
          b"@c\nAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n+\n" + b"I" * 45 + b"\n"
    open(f, "wb").write(cut)
    rc, out, err = run_cli(["sdust", f])
    assert rc == 0 and out == sdust_text(cut) and out.startswith(b"a\t") and b"c\t" not in out


# ---- FASTA framing on the device (cornetto_fasta_split) -------------------------------------------------------------
def fasta_text(rng, n_rec, width=None, crlf=False, blank=0.0, lowc=0.3):
    """FASTA with wrapped lines (width None = one line per record), optional CRLF and blank lines"""
    alpha = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n_rec):
        L = int(rng.choice([0, 1, 59, 60, 61, 120])) if rng.random() < 0.3 else int(rng.integers(1, 3000))
        s = alpha[rng.integers(0, len(alpha), L)].tobytes()
        if L > 100 and rng.random() < lowc:
            a = int(rng.integers(0, L - 60))
            s = s[:a] + b"TTAGGG" * 10 + s[a + 60:]
        head = b">ctg%d" % i + (b" len=%d" % L if rng.random() < 0.5 else b"")
        out.append(head + eol)
        w = width or max(L, 1)
        for k in range(0, L, w):
            out.append(s[k:k + w] + eol)
            if rng.random() < blank:
                out.append(eol)
        if L == 0 and not crlf and rng.random() < 0.5:      # ("\r" alone under an empty record is a base for kseq: not plain)
            out.append(eol)
    return b"".join(out)


def check_fasta(acc, text, final=True):
    recs, used, plain, seqs = acc.fasta_split(text, final=final, want_seqs=True)
    exp, rc = ob.fastx_parse(text)
    got = [(text[int(r["head"]) + 1:int(r["head"]) + 1 + int(r["name_len"])], int(r["len"])) for r in recs]
    assert got == [(n, len(s)) for n, _, s, _ in exp[:len(got)]], (text[:120], got[:3], [(n, len(s)) for n, _, s, _ in exp[:3]])
    # the resident sequences are the records' sequences: telofind + sdust over them equal the oracle's per record
    iv = acc.sdust(seqs, 20, 64)
    want = []
    for i, (_, _, s, _) in enumerate(exp[:len(got)]):
        for v in ob.sdust(np.frombuffer(s, dtype=np.uint8), 20, 64):
            want.append((i, int(v) >> 32, int(v) & 0xFFFFFFFF))
    assert [(int(a), int(b), int(c)) for a, b, c in iv] == want
    hits = acc.telofind(seqs, b"TTAGGG")
    wanth = []
    for i, (_, _, s, _) in enumerate(exp[:len(got)]):
        oh = ob.telofind(np.frombuffer(s, dtype=np.uint8), b"TTAGGG")
        wanth += [(i, int(x["strand"]), int(x["start"]), int(x["end"])) for x in oh]
    assert [tuple(map(int, x)) for x in hits] == wanth
    seqs.close()
    rest, rc2 = ob.fastx_parse(text[used:])
    assert [(n, s) for n, _, s, _ in exp[len(got):]] == [(n, s) for n, _, s, _ in rest]
    return recs, used, plain, exp


def test_fasta_goldens(acc, golden_dir):
    import gzip
    for f in ("probe.fa", "probe_sdust.fa", "probe_selfoverlap.fa"):
        text = open(os.path.join(golden_dir, f), "rb").read()
        if text[:1] == b">":
            check_fasta(acc, text)
    text = gzip.open(os.path.join(golden_dir, "mix.fa.gz")).read()
    recs, used, plain, exp = check_fasta(acc, text)
    assert len(recs) > 0


def test_fasta_random_layouts(acc):
    rng = np.random.default_rng(31)
    for it in range(40):
        text = fasta_text(rng, int(rng.integers(1, 30)), width=[None, 60, 80, 7, 1][it % 5], crlf=(it % 3 == 0), blank=0.1 if it % 4 == 0 else 0.0)
        if it % 7 == 0 and text.endswith(b"\n"):
            text = text[:-1]
        recs, used, plain, exp = check_fasta(acc, text)
        assert plain and used == len(text) and len(recs) == len(exp), it


def test_fasta_text_in_slabs_equals_the_text_in_one_piece(acc):
    """cornetto_text_open / cornetto_text_put (a ring of four pinned slabs, two copy queues, slabs out of order) + cornetto_fasta_split_text
    give the records, the consumed bytes, the plain flag and the resident sequences of cornetto_fasta_split over the same text"""
    rng = np.random.default_rng(33)
    for it in range(12):
        text = fasta_text(rng, int(rng.integers(1, 40)), width=[None, 60, 7][it % 3], crlf=(it % 4 == 0))
        if it == 5:
            text += b"@r1\nACGT\n+\nIIII\n" + fasta_text(rng, 2)
        if it % 5 == 0 and text.endswith(b"\n"):
            text = text[:-1]
        for final in (True, False):
            a = acc.fasta_split(text, final=final, want_seqs=True)
            b = acc.fasta_split_slabs(text, [64, 1000, 4096, 1 << 20][it % 4], final=final, want_seqs=True)
            assert np.array_equal(a[0], b[0]) and a[1:3] == b[1:3], (it, final)
            if len(a[0]):
                assert np.array_equal(acc.sdust(a[3], 20, 64), acc.sdust(b[3], 20, 64))
                assert np.array_equal(acc.telofind(a[3], b"TTAGGG"), acc.telofind(b[3], b"TTAGGG"))
            a[3].close()
            b[3].close()


def test_fasta_not_plain_and_pieces(acc):
    rng = np.random.default_rng(32)
    text = fasta_text(rng, 10, width=60)
    # a FASTQ record in the middle: everything before it is FASTA, the device stops at the record the '@' line follows
    mixed = text + b"@r1\nACGT\n+\nIIII\n" + fasta_text(rng, 3, width=60)
    recs, used, plain, exp = check_fasta(acc, mixed)
    assert not plain and len(recs) == 9
    # "\r"-only first sequence line: kseq keeps it as a base
    odd = b">a\nACGT\n>b\n\r\nAC\r\n>c\nGG\n"
    recs, used, plain, exp = check_fasta(acc, odd)
    assert not plain and len(recs) == 1
    # does not begin with '>'
    recs, used, plain, _ = acc.fasta_split(b"\n>a\nACGT\n")
    assert len(recs) == 0 and used == 0 and not plain
    # pieces: the last record of a piece that is not the last is left for the next call
    text = fasta_text(rng, 200, width=80)
    exp, rc = ob.fastx_parse(text)
    for piece in (1 << 20, 20000, 9000):
        got, pos, pend = [], 0, b""
        while True:
            chunk = text[pos:pos + piece]
            pos += len(chunk)
            final = pos >= len(text)
            buf = pend + chunk
            recs, used, plain, _ = acc.fasta_split(buf, final=final)
            assert plain
            got += [(buf[int(r["head"]) + 1:int(r["head"]) + 1 + int(r["name_len"])], int(r["len"])) for r in recs]
            pend = buf[used:]
            if final:
                break
        assert pend == b"" and got == [(n, len(s)) for n, _, s, _ in exp], piece


def test_fasta_large_wrapped(acc):
    """200 Mbases in 80-column lines: lengths and sampled content against numpy"""
    rng = np.random.default_rng(33)
    lens = [120_000_001, 60_000_000, 19_999_999, 80, 0, 1]
    parts, seqs = [], []
    for i, L in enumerate(lens):
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L, dtype=np.uint8)]
        seqs.append(s)
        k = L // 80 * 80
        m = np.empty((k // 80, 81), dtype=np.uint8)
        m[:, :80] = s[:k].reshape(-1, 80)
        m[:, 80] = 10
        parts += [b">c%d x\n" % i, m.tobytes(), s[k:].tobytes() + (b"\n" if L > k else b"")]
    text = b"".join(parts)
    recs, used, plain, res = acc.fasta_split(text, final=True, want_seqs=True)
    assert plain and used == len(text) and [int(x) for x in recs["len"]] == lens
    hits = acc.telofind(res, b"TTAGGG")
    for i in (0, 2, 3):
        oh = ob.telofind(seqs[i][:2_000_000], b"TTAGGG")
        g = hits[(hits["ctg"] == i) & (hits["end"] <= 2_000_000 - 6)]
        w = oh[oh["end"] <= 2_000_000 - 6]
        assert [(int(x["strand"]), int(x["start"]), int(x["end"])) for x in g] == [(int(x["strand"]), int(x["start"]), int(x["end"])) for x in w]
    res.close()


@pytest.mark.parametrize("piece", [None, "300", "5000", "70000"])
def test_cli_fasta_goldens_any_piece_size(golden_dir, piece):
    """telofind / sdust over FASTA framed on the device (tiny pieces: growth is off, so records that do not fit go to the
    sequential reader) against the reference's stdout"""
    env = {"CORNETTO_FASTQ_PIECE": piece} if piece else None
    for args, exp in ((["sdust", "mix.fa.gz"], "mix.sdust.exp"), (["telofind", "mix.fa.gz"], "mix.telofind.exp"),
                      (["sdust", "probe.fa"], "probe.sdust.exp"), (["telofind", "probe.fa"], "probe.telofind.exp"),
                      (["telofind", "mix.fa.gz", "TTAGGGTTAGGG"], "mix.k12.telofind.exp"), (["sdust", "-w", "32", "-t", "10", "mix.fa.gz"], "mix.w32t10.sdust.exp")):
        a = [os.path.join(golden_dir, x) if os.path.exists(os.path.join(golden_dir, x)) else x for x in args]
        rc, out, err = run_cli(a, env=env)
        assert rc == 0, err.decode()
        assert out == open(os.path.join(golden_dir, exp), "rb").read(), (args, piece)


def test_cli_fasta_then_fastq_in_one_file_and_stdin(tmp_path):
    rng = np.random.default_rng(41)
    text = fasta_text(rng, 12, width=60) + lowcomplex_fastx(rng, 10, True) + b"\n" + fasta_text(rng, 5, width=None, crlf=True)
    f = str(tmp_path / "mixed.fa")
    open(f, "wb").write(text)
    want = sdust_text(text)
    assert len(want) > 100
    for piece in ("2000", None):
        env = {"CORNETTO_FASTQ_PIECE": piece} if piece else None
        rc, out, err = run_cli(["sdust", f], env=env)
        assert rc == 0 and out == want, piece
        rc, out, err = run_cli(["sdust", "-"], data=text, env=env)
        assert rc == 0 and out == want, piece
    rc, out, err = run_cli(["sdust", f], env={"CORNETTO_FASTQ_SPLIT": "host"})
    assert rc == 0 and out == want


@pytest.mark.parametrize("big_at", [None, 0, 7, 19])
def test_cli_small_first_piece_then_full_pieces_read_ahead(tmp_path, big_at):
    """round 5: with read-ahead the first pinned piece of an uncompressed FASTA file is small (CORNETTO_CLI_FIRST_MB, here 1 MiB of a
    ~7 MB file) and the full-size buffers are made by the read-ahead thread; the small buffer comes back as the second buffer and is
    made again at the full size; a record larger than the first piece (at the start, in the middle, at the end) grows it.  Same bytes as
    without read-ahead and as the reference's sdust over the same records."""
    rng = np.random.default_rng(99 if big_at is None else big_at)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for i in range(20):
        L = 1_600_000 if big_at == i else int(rng.integers(150_000, 400_000))
        sq = acgt[rng.integers(0, 4, L)].copy()
        for _ in range(L // 20_000):
            a = int(rng.integers(0, L - 300))
            sq[a:a + 240] = np.frombuffer(b"TTAGGG" * 40, dtype=np.uint8)
        body = sq.tobytes()
        w = 80 if i % 3 else 0
        recs.append(b">ctg%d some text\n" % i + (b"".join(body[k:k + w] + b"\n" for k in range(0, L, w)) if w else body + b"\n"))
    text = b"".join(recs)
    f = str(tmp_path / "big.fa")
    open(f, "wb").write(text)
    want = sdust_text(text)
    assert len(want) > 2000
    rc0, out0, err0 = run_cli(["sdust", f], env={"CORNETTO_CLI_AHEAD": "0"})
    assert rc0 == 0 and out0 == want
    for first in ("1", "2", "64"):
        rc, out, err = run_cli(["sdust", f], env={"CORNETTO_CLI_FIRST_MB": first})
        assert rc == 0 and out == want, (first, err[-300:])
    rc, out, err = run_cli(["telofind", f], env={"CORNETTO_CLI_FIRST_MB": "1"})
    rc1, out1, _ = run_cli(["telofind", f], env={"CORNETTO_CLI_AHEAD": "0"})
    assert rc == 0 and rc1 == 0 and out == out1 and len(out) > 1000


def test_cli_pieces_grow_until_the_largest_record_fits(golden_dir):
    env = {"CORNETTO_FASTQ_PIECE": "128", "CORNETTO_FASTQ_GROW": "1"}
    for args, exp in ((["sdust", "mix.fa.gz"], "mix.sdust.exp"), (["telofind", "probe.fa"], "probe.telofind.exp"), (["sdust", "reads.fq"], "reads.sdust.exp")):
        rc, out, err = run_cli([args[0], os.path.join(golden_dir, args[1])], env=env)
        assert rc == 0 and out == open(os.path.join(golden_dir, exp), "rb").read(), args
