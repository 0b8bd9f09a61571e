#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference binary.

Run in the build container only (needs oracle/_ref/cornetto, built by `make -f oracle/ref.mk` from
/root/reference).  Inputs are synthetic (seeded numpy) or hand-written probes; expected outputs are the
reference's stdout, byte for byte.  Inputs AND outputs are committed, so nothing here has to be
re-run on the GPU box and the numpy bit-stream need not be stable across versions.

Also copies the DATA files of the reference's own bigenough test (test/bigenough/hg002-cornetto-E_3,
used by test/test.sh:33-39) — fixtures, not source.

    python tests/golden/make_golden.py
"""
import gzip
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "cornetto")
REFSRC = "/root/reference"


def run(args, out_path, stdin=None, expect=0):
    """run the reference, store stdout at out_path, return exit status (expect: the status it must have; -6 = SIGABRT)"""
    with open(out_path, "wb") as fo:
        p = subprocess.run([REF] + args, stdout=fo, stderr=subprocess.DEVNULL, stdin=stdin)
    assert p.returncode == expect, (args, p.returncode)
    return p.returncode


def wrap(seq: bytes, width: int, eol: bytes = b"\n") -> bytes:
    if width <= 0:
        return seq + eol
    return b"".join(seq[i:i + width] + eol for i in range(0, len(seq), width)) if seq else eol


def rand_bases(rng, n):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].tobytes()


def plant(buf: bytearray, pos: int, s: bytes):
    s = s[: max(0, len(buf) - pos)]
    buf[pos:pos + len(s)] = s


def make_mix_fasta(rng) -> bytes:
    out = []
    # ctgA: 600 kb, 80-col, telomere arrays at both ends + one feature every 20 kb
    n = 600_000
    a = bytearray(rand_bases(rng, n))
    plant(a, 0, b"CCCTAA" * 300)
    plant(a, n - 1500, b"TTAGGG" * 250)
    kinds = 0
    for pos in range(20_000, n - 20_000, 20_000):
        k = kinds % 9
        kinds += 1
        if k == 0:
            plant(a, pos, b"TTAGGG" * int(rng.integers(3, 81)))
        elif k == 1:
            plant(a, pos, bytes([b"ACGT"[int(rng.integers(0, 4))]]) * int(rng.integers(10, 301)))
        elif k == 2:
            plant(a, pos, rand_bases(rng, 2) * int(rng.integers(10, 201)))
        elif k == 3:
            plant(a, pos, b"N" * int(rng.integers(1, 501)))
        elif k == 4:
            L = 500
            a[pos:pos + L] = bytes(a[pos:pos + L]).lower()
        elif k == 5:
            plant(a, pos, rand_bases(rng, 3) * int(rng.integers(10, 120)))
        elif k == 6:
            plant(a, pos, b"CCCTAA" * int(rng.integers(3, 81)) + b"RYKM" + b"ttaggg" * 7)
        elif k == 7:
            # low complexity directly against an N run (the stale-window quirk)
            plant(a, pos, b"AC" * 16 + b"NNNN" + b"AC" * 13 + b"N" + b"GGGGGGGGGGGGGGGGGGGGGGGGGGGG")
        else:
            plant(a, pos, b"TTAGGG" * 40 + b"TTAGG" + b"TTAGGG" * 3 + b"CCCTAA" * 5)
    out.append(b">ctgA some description\n" + wrap(bytes(a), 80))
    # ctgB: 300 kb, single line, with an N-dense stretch (short ACGT runs between Ns)
    n = 300_000
    b = bytearray(rand_bases(rng, n))
    pos = 100_000
    while pos < 130_000:
        run_len = int(rng.integers(1, 12))
        gap = int(rng.integers(1, 6))
        plant(b, pos + run_len, b"N" * gap)
        pos += run_len + gap
    # low-complexity mosaic touching the N-dense stretch
    plant(b, 129_000, (b"AAT" * 30 + b"N" + b"AT" * 40 + b"NN" + b"A" * 70) * 4)
    plant(b, n - 600, b"TTAGGG" * 100)
    out.append(b">ctgB\n" + wrap(bytes(b), 0))
    # ctgC: empty record
    out.append(b">ctgC empty\n\n")
    # ctgD: 5 kb with CRLF line ends
    d = bytearray(rand_bases(rng, 5_000))
    plant(d, 0, b"CCCTAA" * 100)
    out.append(b">ctgD\tcrlf\r\n" + wrap(bytes(d), 60, b"\r\n"))
    # ctgE: 150 kb adversarial STR mosaic (about half masked by sdust)
    parts = []
    tot = 0
    while tot < 150_000:
        k = int(rng.integers(0, 4))
        if k == 0:
            p = rand_bases(rng, int(rng.integers(20, 200)))
        elif k == 1:
            p = rand_bases(rng, int(rng.integers(1, 7))) * int(rng.integers(4, 40))
        elif k == 2:
            p = bytes([b"ACGT"[int(rng.integers(0, 4))]]) * int(rng.integers(5, 90))
        else:
            p = b"N" * int(rng.integers(1, 4))
        parts.append(p)
        tot += len(p)
    e = b"".join(parts)[:150_000]
    out.append(b">ctgE\n" + wrap(e, 70))
    # short records
    out.append(b">ctgF\n" + rand_bases(rng, 63) + b"\n")
    out.append(b">ctgG\nAC\n")
    out.append(b">ctgH\nTTAGGG\n")
    out.append(b">ctgI\n" + b"A" * 64 + b"\n")
    out.append(b">ctgJ lower\n" + wrap((b"ttaggg" * 500 + rand_bases(rng, 3000).lower() + b"ccctaa" * 400), 100))
    return b"".join(out)


BG_CTGS = [("ptg000001l", 120), ("ptg000002l", 2500), ("ptg000003l", 2551), ("ptg000004l", 12_000),
           ("ptg000005l", 40_000), ("ptg000006l", 2450), ("ptg000007l", 10_000)]
# a second pair for `-i` larger than `-w` (src/boringbits_main.c:338-369: sparse windows; the last one must reach the contig's end or
# the reference aborts): every length is <= 64 or in (1000 k, 1000 k + 64], so `-w 64 -i 1000` gets through; 2033 = 1982 + 51 is the
# contig of exactly w + 51 for `-w 1982 -i 7` (w % inc = 1: the head sums)
SPARSE_CTGS = [("utg000001l", 15_001), ("utg000002l", 3064), ("utg000003l", 40), ("utg000004l", 64), ("utg000005l", 1001),
               ("utg000006l", 2033), ("utg000007l", 7050)]


def make_bedgraphs(rng, ctgs=BG_CTGS, step=4000, first=1500):
    """two lock-step per-base bedgraphs: lines `name\\tpos\\tpos+1\\tdepth`"""
    tot_lines, mq_lines = [], []
    for name, n in ctgs:
        base = rng.poisson(30, size=(n + 999) // 1000).repeat(1000)[:n]
        depth = base + rng.integers(-2, 3, size=n)
        depth = np.clip(depth, 0, None)
        mq = depth.copy()
        for pos in range(first, n, step):
            L = int(rng.integers(200, 1500))
            k = (pos // step) % 5
            if k == 0:
                depth[pos:pos + L] //= 5
                mq[pos:pos + L] = depth[pos:pos + L]
            elif k == 1:
                depth[pos:pos + L] *= 3
                mq[pos:pos + L] = depth[pos:pos + L]
            elif k == 2:
                mq[pos:pos + L] //= 4
            elif k == 3:
                depth[pos:pos + L] = 0
                mq[pos:pos + L] = 0
            else:
                depth[pos:pos + 3] = 70_000          # > 65535: clamped by the reader (3 positions)
                mq[pos:pos + 3] = 70_005
                mq[pos + 3:pos + L] = depth[pos + 3:pos + L] + 5   # mq > depth
        for i in range(n):
            tot_lines.append("%s\t%d\t%d\t%d\n" % (name, i, i + 1, depth[i]))
            mq_lines.append("%s\t%d\t%d\t%d\n" % (name, i, i + 1, mq[i]))
    return "".join(tot_lines).encode(), "".join(mq_lines).encode()


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -f oracle/ref.mk")
    rng = np.random.default_rng(20260807)
    tmp = os.path.join(HERE, "_tmp")
    os.makedirs(tmp, exist_ok=True)
    os.chdir(HERE)

    # ---------------- probes ----------------
    with open("probe.fa", "wb") as f:
        f.write(b">c1 desc here\nttagggTTAGGGTTAGGGNTTAGGGTTAGG\nGTTAGGGCCCTAACCCTAAxCCCTAA\n>c2\n\n>c3\nACGT\n"
                b">c4\nTTAGGGTTAGGGTTAGGG\n")
    with open("probe_selfoverlap.fa", "wb") as f:
        f.write(b">x\nAAAAAAAAAAAGAAAAT\n>y\nACACACACAGACACACATTACACACAC\n>z\nAAAAAAAAAAAAAAAAAAAAAAAA\n")
    with open("probe_sdust.fa", "wb") as f:
        f.write(b">s1\nAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAACGTACGATCGATCGTAGCTAGCTAGCTAGCATCGATCGACTAGCTAGCATCGATC"
                b"AGCATCGACTAGCATCAGCTACGACTACGACTACGAGCGAGCGACGGCATTATATATATATATATATATATATATATATATATATATATAT\n"
                b">s2\nacacacacacacacacacacacacacacacacNNNNacacacacacacacacacacacacac\n>s3\nNNNN\n>s4\nAC\n>s5\n\n")
    with open("probe.telomere", "wb") as f:
        f.write(b"c1\t2500\t0\t0\t600\t600\nc1\t2500\t0\t2300\t2500\t200\nc1\t2500\t1\t900\t1000\t100\n"
                b"c2\t700\t0\t0\t300\t300\nc3\t1000\t1\t500\t1000\t500\nc4\t1000\t0\t0\t1000\t1000\n"
                b"c5\t999\t0\t100\t600\t500\nc6\t1200\t0\t700\t1200\t500\nc1\t2500\t0\t0\t100\t100\n")
    run(["telofind", "probe.fa"], "probe.telofind.exp")
    run(["fa2bed", "probe.fa"], "probe.fa2bed.exp")
    run(["sdust", "probe.fa"], "probe.sdust.exp")
    run(["telofind", "probe_selfoverlap.fa", "AAAA"], "probe_selfoverlap.AAAA.telofind.exp")
    run(["telofind", "probe_selfoverlap.fa", "ACAC"], "probe_selfoverlap.ACAC.telofind.exp")
    run(["telofind", "probe_selfoverlap.fa", "ACACA"], "probe_selfoverlap.ACACA.telofind.exp")
    run(["sdust", "probe_sdust.fa"], "probe_sdust.sdust.exp")
    run(["sdust", "-w", "32", "-t", "10", "probe_sdust.fa"], "probe_sdust.w32t10.sdust.exp")
    run(["telowin", "probe.telomere", "99.9", "0.4"], "probe.telowin.exp")
    run(["telowin", "probe.telomere", "100", "0.5"], "probe.i100t05.telowin.exp")
    run(["telowin", "probe.telomere", "95"], "probe.i95.telowin.exp")

    # ---------------- mixed FASTA ----------------
    fa = make_mix_fasta(rng)
    with gzip.GzipFile("mix.fa.gz", "wb", mtime=0) as f:
        f.write(fa)
    mix = os.path.join(tmp, "mix.fa")
    with open(mix, "wb") as f:
        f.write(fa)
    run(["telofind", mix], "mix.telofind.exp")
    run(["telofind", "mix.fa.gz"], os.path.join(tmp, "mix.gz.telofind"))
    assert open("mix.telofind.exp", "rb").read() == open(os.path.join(tmp, "mix.gz.telofind"), "rb").read()
    run(["telofind", mix, "ttaggg"], "mix.lower_motif.telofind.exp")
    run(["telofind", mix, "TTAGGGTTAGGG"], "mix.k12.telofind.exp")
    run(["telofind", mix, "TTAGGG" * 6], "mix.k36.telofind.exp")            # longer than the 32-byte automaton of the device path
    run(["telofind", mix, "GGGTTA" * 11 + "G"], "mix.k67.telofind.exp")
    run(["telofind", mix, "AAAA"], "mix.AAAA.telofind.exp")
    run(["telofind", mix, "GNG"], "mix.GNG.telofind.exp")
    run(["telowin", "mix.telofind.exp", "99.9", "0.4"], "mix.telowin.exp")
    run(["telowin", "mix.telofind.exp", "99.9", "0.1"], "mix.t01.telowin.exp")
    run(["fa2bed", mix], "mix.fa2bed.exp")
    run(["sdust", mix], "mix.sdust.exp")
    run(["sdust", "-w", "32", "-t", "10", mix], "mix.w32t10.sdust.exp")
    run(["sdust", "-w", "100", "-t", "25", mix], "mix.w100t25.sdust.exp")
    run(["sdust", "-w", "16", "-t", "30", mix], "mix.w16t30.sdust.exp")
    run(["sdust", "-t", "5", mix], "mix.t5.sdust.exp")

    # ---------------- FASTQ for seq ----------------
    recs = []
    for i in range(40):
        L = int(rng.integers(0, 400))
        s = rand_bases(rng, L)
        q = bytes((rng.integers(3, 41, size=L) + 33).astype(np.uint8))
        cm = b"" if i % 3 == 0 else b" runid=abc ch=%d" % (i * 7)
        recs.append(b"@read%d" % i + cm + b"\n" + s + b"\n+\n" + q + b"\n")
    with open("reads.fq", "wb") as f:
        f.write(b"".join(recs))
    run(["seq", "-m", "100", "reads.fq"], "reads.m100.seq.exp")
    run(["seq", "reads.fq"], "reads.default.seq.exp")
    run(["sdust", "reads.fq"], "reads.sdust.exp")
    run(["fa2bed", "reads.fq"], "reads.fa2bed.exp")

    # ---------------- bedgraphs ----------------
    tot, mq = make_bedgraphs(rng)
    for name, data in (("cov-total.bg", tot), ("cov-mq20.bg", mq)):
        with gzip.GzipFile(name + ".gz", "wb", mtime=0) as f:
            f.write(data)
        with open(os.path.join(tmp, name), "wb") as f:
            f.write(data)
    t, q = os.path.join(tmp, "cov-total.bg"), os.path.join(tmp, "cov-mq20.bg")
    # the two option sets of the reference's own test/test.sh:25,29 + defaults + odd sizes
    run(["boringbits", t, "-q", q, "-m", "10000", "-e", "1000", "-L", "0.6", "-Q", "0.6", "-H", "1.6"], "bg.boring_t1.exp")
    run(["noboringbits", "-H", "2.5", "-L", "0.5", "-Q", "0.5", t, "-q", q, "-m", "10000", "-e", "1000"], "bg.fun_t2.exp")
    run(["noboringbits", t, "-q", q], "bg.fun_default.exp")
    run(["boringbits", t, "-q", q], "bg.boring_default.exp")
    run(["noboringbits", t, "-q", q, "-w", "300", "-i", "7", "-L", "0.33", "-H", "1.45", "-Q", "0.9", "-m", "5000", "-e", "500"], "bg.fun_w300i7.exp")
    run(["boringbits", t, "-q", q, "-w", "300", "-i", "7", "-L", "0.33", "-H", "1.45", "-Q", "0.9", "-m", "5000", "-e", "500"], "bg.boring_w300i7.exp")
    run(["noboringbits", t, "-q", q, "-w", "1000", "-i", "1000", "-m", "2000", "-e", "10000"], "bg.fun_w1000i1000.exp")
    # the options the reference accepts and ignores (src/boringbits_main.c:590-632): same bytes as the defaults
    run(["noboringbits", t, "-q", q, "-t", "4", "-K", "10", "-B", "1M", "-o", os.path.join(tmp, "x"), "--debug-break", "1", "--profile-cpu", "yes", "--accel=yes"],
        os.path.join(tmp, "ignored.out"))
    assert open(os.path.join(tmp, "ignored.out"), "rb").read() == open("bg.fun_default.exp", "rb").read()
    # -i larger than -w (:338-369): disjoint windows [j*inc, min(j*inc + w, len)); every contig's last window must be non-empty and end at
    # the contig's end, else assert(st<end) (:353) raises SIGABRT before anything is printed (stdout is never flushed)
    run(["noboringbits", t, "-q", q, "-w", "300", "-i", "301", "-m", "5000", "-e", "500"], "bg.fun_w300i301.exp")
    run(["boringbits", t, "-q", q, "-w", "300", "-i", "301", "-m", "5000", "-e", "500", "-L", "0.33", "-H", "1.45"], "bg.boring_w300i301.exp")
    run(["noboringbits", t, "-q", q, "-w", "300", "-i", "350"], "bg.abort_w300i350.exp", expect=-6)      # 2450 = 7 x 350: window 7 is empty
    run(["boringbits", t, "-q", q, "-w", "64", "-i", "1000", "-m", "100"], "bg.abort_w64i1000.exp", expect=-6)   # the first contig (120) already; -m does not save it
    rng2 = np.random.default_rng(20260808)      # (its own stream: nothing above moves when this part changes)
    tot2, mq2 = make_bedgraphs(rng2, SPARSE_CTGS, step=1700, first=300)
    for name, data in (("sparse-total.bg", tot2), ("sparse-mq20.bg", mq2)):
        with gzip.GzipFile(name + ".gz", "wb", mtime=0) as f:
            f.write(data)
        with open(os.path.join(tmp, name), "wb") as f:
            f.write(data)
    t2, q2 = os.path.join(tmp, "sparse-total.bg"), os.path.join(tmp, "sparse-mq20.bg")
    run(["noboringbits", t2, "-q", q2, "-w", "64", "-i", "1000", "-m", "1000", "-e", "200"], "sparse.fun_w64i1000.exp")
    run(["boringbits", t2, "-q", q2, "-w", "64", "-i", "1000", "-m", "1000", "-e", "200", "-L", "0.2", "-H", "3"], "sparse.boring_w64i1000.exp")
    run(["noboringbits", t2, "-q", q2, "-w", "64", "-i", "1000", "-m", "100000"], "sparse.fun_w64i1000_short.exp")   # every contig below -m: get_regs ran all the same
    run(["noboringbits", t2, "-q", q2, "-w", "1982", "-i", "7", "-e", "5000", "-m", "1000"], "sparse.fun_w1982i7e5000.exp")   # -e beyond every contig, w % inc = 1
    run(["boringbits", t2, "-q", q2, "-w", "1982", "-i", "7", "-e", "5", "-m", "1000", "-L", "0.2", "-H", "3"], "sparse.boring_w1982i7.exp")
    run(["noboringbits", t2, "-q", q2], "sparse.fun_default.exp")
    run(["noboringbits", t2, "-q", q2, "-w", "64", "-i", "999"], "sparse.abort_w64i999.exp", expect=-6)

    # ---------------- bigenough ----------------
    dst = os.path.join(HERE, "bigenough")
    os.makedirs(dst, exist_ok=True)
    src = os.path.join(REFSRC, "test", "bigenough", "hg002-cornetto-E_3")
    for fn in os.listdir(src):
        shutil.copyfile(os.path.join(src, fn), os.path.join(dst, fn))
        os.chmod(os.path.join(dst, fn), 0o644)
    # int32-overflow cases
    with open(os.path.join(dst, "ovf_chroms.bed"), "w") as f:
        f.write("big1\t0\t90709979\nbig2\t0\t61364156\nsmall\t0\t1000\nhalf\t0\t2000\nbig3\t0\t242000000\n")
    with open(os.path.join(dst, "ovf_in.bed"), "w") as f:
        f.write("big1\t100\t3000100\nbig2\t5\t10\nsmall\t0\t500\nhalf\t0\t1001\nbig1\t5000000\t5000010\n"
                "big3\t0\t121000000\nsmall\t400\t450\nbig3\t130000000\t130000001\n")
    for T in ("50", "0", "100", "33"):
        run(["bigenough", "-T", T, os.path.join(dst, "ovf_chroms.bed"), os.path.join(dst, "ovf_in.bed"),
             "-r", os.path.join(dst, "ovf_T%s.csv.exp" % T)], os.path.join(dst, "ovf_T%s.bed.exp" % T))

    shutil.rmtree(tmp)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
