#!/usr/bin/env python3
"""development aid: randomised differential run of the CLI on the DEVICE path against the unmodified reference binary (oracle/_ref/cornetto, which
travels with the snapshot): stdout and exit status, byte for byte, with seeds the test suite does not use.
  fasta   telofind / sdust (-w / -t varied) / fa2bed / seq on random FASTA and FASTQ text: wrapped and unwrapped records, CRLF, lower case, N runs and
          IUPAC letters, empty records, a missing last newline, planted telomere arrays and low-complexity runs, gzip input, stdin
  panel   (no)boringbits on random per-base bedgraph pairs: -w / -i smaller, equal, larger than each other, -m, -e, -L, -H, -Q; contig lengths at and around
          multiples of -i; the command lines on which the reference dies of its assert (SIGABRT); malformed lines (the reference's five checks: exit 1);
          one device, contigs dealt to several handles, the text itself cut into shares
  telo    telowin (identity and threshold varied) and telobreaks on the reference's own telofind / sdust / fa2bed outputs for such a FASTA (scripts/telostats.sh)
  bigenough  (host only) random assembly bed / boring-bits bed / -T / -r, lengths up to and beyond 2^31, malformed entries
  khash   telobreaks on synthetic tables of up to 30 000 contigs that all print: the reference's hash-table order
   python tools/fuzz_cli.py [fasta|telo|panel|bigenough|khash|all] [first_seed] [n_seeds]"""
import gzip
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.environ.get("CORNETTO_FUZZ_CLI") or os.path.join(ROOT, "cornetto_amd", "cornetto")      # (e.g. cornetto_amd/cornetto_asan: make -C cornetto_amd asan=1)
REF = os.path.join(ROOT, "oracle", "_ref", "cornetto")


def run(binary, args, env=None, data=None):
    e = dict(os.environ)
    e.update(env or {})
    try:
        p = subprocess.run([binary] + args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=60)
    except subprocess.TimeoutExpired:
        return "timeout", b"", b"timeout"
    return p.returncode, p.stdout, p.stderr


def rand_seq(rnd, n):
    kind = rnd.random()
    alpha = "ACGT" if kind < 0.6 else "ACGTacgt" if kind < 0.8 else "ACGTNacgtnRYKM"
    s = [rnd.choice(alpha) for _ in range(n)]
    for _ in range(rnd.randint(0, max(1, n // 300))):
        if n < 20:
            break
        p = rnd.randrange(n)
        unit = rnd.choice(["TTAGGG", "CCCTAA", "A", "AT", "CAG", "N", "ttaggg", "GGAAT", "TTAGGGTTAGGC", "n"])
        rep = (unit * rnd.randint(1, 120))[: n - p]
        s[p:p + len(rep)] = list(rep)
    return "".join(s)


def fasta_text(rnd, max_rec=7, max_len=40000, dirty=False):
    fastq = rnd.random() < 0.25
    eol = "\r\n" if rnd.random() < 0.15 else "\n"
    out = []
    style = rnd.randrange(4)
    for r in range(rnd.randint(1, max_rec)):
        n = rnd.choice([0, 1, 2, 5, 63, 64, 65, 200, 1000, 5000, 20000, 70000]) if rnd.random() < 0.5 else rnd.randint(0, max_len)
        n = min(n, max(max_len, 70000) if max_rec <= 7 else max_len)
        seq = rand_seq(rnd, n)
        name = ("ctg%d" % r, "ptg%06dl" % r, "h%dtg%06dl" % (1 + r % 2, r), "chr%d_MATERNAL|arrow" % r)[style] + (" some comment" if rnd.random() < 0.3 else "")
        if fastq:
            out.append("@" + name + eol + seq + eol + "+" + eol + "".join(rnd.choice("!#5I~") for _ in range(n)) + eol)
        else:
            wrap = rnd.choice([0, 0, 60, 70, 7])
            body = seq if not wrap else eol.join(seq[i:i + wrap] for i in range(0, len(seq), wrap))
            out.append(">" + name + eol + body + (eol if body or rnd.random() < 0.7 else ""))
    text = "".join(out)
    if rnd.random() < 0.2 and text.endswith(eol):
        text = text[: -len(eol)]
    if dirty and rnd.random() < 0.5:
        # what kseq_read() (src/kseq.h:184-224) makes of text that is not well-formed: lines dropped, doubled, blank, header-like lines inside a record,
        # a record without its '+', a cut-off end
        lines = text.split(eol)
        for _ in range(rnd.randint(1, 5)):
            if not lines:
                break
            k = rnd.randrange(len(lines))
            m = rnd.randrange(8)
            if m == 0:
                del lines[k]
            elif m == 1:
                lines.insert(k, lines[k])
            elif m == 2:
                lines.insert(k, "")
            elif m == 3:
                lines.insert(k, rnd.choice([">x", "@y z", "+", "+x", ">", "@"]))
            elif m == 4:
                lines[k] = rnd.choice([">", "@", "+", " ", "\t"]) + lines[k]
            elif m == 5:
                lines[k] = lines[k] + rnd.choice([" ", "\t", "\r", " x"])
            elif m == 6:
                lines[k] = lines[k][: len(lines[k]) // 2]
            else:
                lines = lines[: k + 1]
        text = eol.join(lines)
        if rnd.random() < 0.5:
            text = text[: rnd.randint(0, len(text))]
    return text.encode()


def fuzz_fasta(seed, tmp):
    rnd = random.Random(seed)
    text = fasta_text(rnd, dirty=True)
    path = os.path.join(tmp, "f.fa")
    gz = rnd.random() < 0.2
    if gz:
        path += ".gz"
        with gzip.open(path, "wb") as f:
            f.write(text)
    else:
        with open(path, "wb") as f:
            f.write(text)
    sub = rnd.choice(["telofind", "sdust", "sdust", "fa2bed", "seq"])
    args, data = [sub], None
    motif = None
    if sub == "telofind" and rnd.random() < 0.6:
        # the optional second argument (src/find_telomere.c:89-91): any string — with and without a border, longer than the automaton's 32 bytes, lower
        # case (never found: the sequence is upper-cased, the motif is not), letters that are no bases
        motif = rnd.choice(["TTAGGG", "CCCTAA", "TTTAGGG", "TTAGGGTTAGGG", "AAAA", "ACACA", "A", "AC", "GATC", "ttaggg", "TTANGG", "N", "NN", "GGAAT", "CATTC",
                            "ACGT" * 9, "TTAGGG" * 7, "TA" * 20, "R", "TTAGGGT", "AT", "CAG"])
    if sub == "sdust":
        if rnd.random() < 0.6:
            args += ["-w", str(rnd.choice([64, 40, 30, 16, 8, 66, 100, 200, 5, 4, 3])), "-t", str(rnd.choice([20, 25, 10, 5, 30, 2, 1, 0, 1000]))]
    elif sub == "seq":
        if rnd.random() < 0.7:
            args += ["-m", str(rnd.choice([0, 1, 100, 1000, 10000]))]
    if rnd.random() < 0.15 and not gz:
        args.append("-")
        data = text
    else:
        args.append(path)
    if motif is not None:
        args.append(motif)
    rr = run(REF, args, data=data)
    env = {"CORNETTO_DEVICES": "0,0"} if rnd.random() < 0.2 else {}
    # the streaming paths of the CLI, with sizes that make a small text many pieces: the piece loop (a record cut by a piece's end, a piece without a
    # record start), the read-ahead and the whole-text path on and off, the sequential reader instead of the device's framing, batches of a few records
    if rnd.random() < 0.6:
        for key, vals in (("CORNETTO_FASTQ_PIECE", ["64", "100", "517", "3000", "65536"]), ("CORNETTO_BATCH_BASES", ["1", "700", "50000"]), ("CORNETTO_CLI_WHOLE", ["0", "1"]),
                          ("CORNETTO_CLI_AHEAD", ["0", "1"]), ("CORNETTO_READ_THREADS", ["1", "3"]), ("CORNETTO_CLI_MMAP", ["0", "1"]), ("CORNETTO_FASTQ_SPLIT", ["host", "device"]),
                          ("CORNETTO_FASTQ_GROW", ["0", "1"])):
            if rnd.random() < 0.35:
                env[key] = rnd.choice(vals)
    gg = run(CLI, args, env, data=data)
    # (seq on FASTA prints "(null)" for the absent quality in the reference: undefined behaviour, SURVEY appendix A-5 — not compared)
    if sub == "seq" and b"(null)" in rr[1]:
        return True, None
    if sub == "seq":
        # (the same for a record that ends at its sequence — a cut-off file —: kseq_read() returns it without a quality, src/kseq.h:213, and the reference
        # prints what its quality buffer still holds from the record before)
        ls = rr[1].split(b"\n")
        if any(len(ls[i + 1]) != len(ls[i + 3]) for i in range(0, len(ls) - 3, 4)):
            return True, None
    if rr[0] in ("timeout", -11, -6):                    # (the reference itself hangs or dies: nothing to compare with)
        return True, None
    ok = (gg[0], gg[1]) == (rr[0], rr[1])
    return ok, None if ok else (args, env, gz, len(text), gg[0], rr[0], len(gg[1]), len(rr[1]), gg[2][-300:])


def fuzz_telo(seed, tmp):
    """the pipeline of scripts/telostats.sh on a random FASTA: the reference's own telofind / sdust / fa2bed outputs are the inputs of telowin and telobreaks"""
    rnd = random.Random(seed)
    many = rnd.random() < 0.4
    text = fasta_text(rnd, 60, 6000) if many else fasta_text(rnd)      # (many short contigs: the reference's hash tables grow, its print order is their bucket order)
    path = os.path.join(tmp, "p.fa")
    open(path, "wb").write(text)
    tel = run(REF, ["telofind", path])[1]
    sd = run(REF, ["sdust", path])[1]
    bed = run(REF, ["fa2bed", path])[1]
    lens = b"".join(f.split(b"\t")[0] + b"\t" + f.split(b"\t")[2] + b"\n" for f in bed.splitlines() if f.count(b"\t") >= 2)
    if rnd.random() < 0.3:                               # (any order of the lines: the reference's hash tables do not care)
        ls = sd.splitlines(True)
        rnd.shuffle(ls)
        sd = b"".join(ls)
    if rnd.random() < 0.4 and tel:
        # the telofind rows as telowin reads them (src/telomere_windows.c:65-80: six %s per line, atoi on three of them, a new scaffold whenever the
        # name changes): rows in any order inside a contig, twice, with blanks for tabs, a seventh column, empty and reversed spans, signs and zeros
        # in the numbers, a contig's rows in two places (the name comes back: a second scaffold of that name).  Every row keeps six columns and
        # stays inside its contig: fewer columns and spans beyond the length are stale stack and heap writes in the reference.
        rows = [r.split(b"\t") for r in tel.splitlines() if r.count(b"\t") >= 5]
        blocks, cur = [], None
        for r in rows:
            if cur is None or cur[0][0] != r[0]:
                cur = []
                blocks.append(cur)
            cur.append(r)
        for b_ in blocks:
            if rnd.random() < 0.5:
                rnd.shuffle(b_)
            for r in list(b_):
                k = rnd.random()
                if k < 0.05:
                    b_.append(list(r))
                elif k < 0.10:
                    r[3], r[4] = r[4], r[3]
                elif k < 0.15:
                    r[4] = r[3]
                elif k < 0.20:
                    r[3] = b"+" + r[3]
                    r[4] = b"00" + r[4]
                elif k < 0.25:
                    r.append(b"extra")
        if len(blocks) > 1 and rnd.random() < 0.3:
            k = rnd.randrange(len(blocks))
            if len(blocks[k]) > 1:
                h = len(blocks[k]) // 2
                blocks.append(blocks[k][h:])
                blocks[k] = blocks[k][:h]
        sep = rnd.choice([b"\t", b" ", b"  \t "])
        tel_w = b"".join(sep.join(r) + b"\n" for b_ in blocks for r in b_)
    else:
        tel_w = tel
    pt, ps, pl = os.path.join(tmp, "p.telomere"), os.path.join(tmp, "p.sdust"), os.path.join(tmp, "p.lens")
    is_win = rnd.random() < 0.5
    if not is_win and rnd.random() < 0.5:
        # telobreaks (src/telomere_breaks.c:60-128): the lens file in another order, a name twice, names that are missing (their rows are dropped), the
        # telomere rows in any order
        ll = lens.splitlines(True)
        if rnd.random() < 0.5:
            rnd.shuffle(ll)
        if ll and rnd.random() < 0.3:
            ll.insert(rnd.randrange(len(ll) + 1), rnd.choice(ll))
        if len(ll) > 1 and rnd.random() < 0.4:
            del ll[rnd.randrange(len(ll))]
        lens = b"".join(ll)
        tl = tel.splitlines(True)
        if rnd.random() < 0.5:
            rnd.shuffle(tl)
        tel = b"".join(tl)
    open(pt, "wb").write(tel_w if is_win else tel)
    open(ps, "wb").write(sd)
    open(pl, "wb").write(lens)
    if is_win:
        args = ["telowin", pt, rnd.choice(["99.9", "90", "100", "50", "1e2", "99.99999"]), rnd.choice(["0.4", "0.1", "0.8", "0.01", "1", "0"])]
        if rnd.random() < 0.3:
            args = args[:3]                              # (the threshold is optional: src/telomere_windows.c:54-56)
    else:
        args = ["telobreaks", pl, ps, pt]
    rr = run(REF, args)
    gg = run(CLI, args)
    # (an sdust interval that ends beyond its contig — sdust prints them at a contig's end — is an unchecked write beyond the reference's bitset,
    # src/telomere_breaks.c:85: most of the time nothing reads it, sometimes the allocator notices (SIGABRT) or the page ends (SIGSEGV): not compared)
    if args[0] == "telobreaks" and rr[0] in (-6, -11):
        return True, None
    ok = (gg[0], gg[1]) == (rr[0], rr[1])
    return ok, None if ok else (args[0], args[2:] if args[0] == "telowin" else "", len(text), gg[0], rr[0], len(gg[1]), len(rr[1]), gg[2][-300:])


def fuzz_khash(seed, tmp):
    """telobreaks on synthetic tables with MANY contigs whose every telomere row lies inside a low-complexity run: every contig prints, in the bucket
    order of the reference's hash table (khash 0.2.8: src/khash.h) after its resizes — names of several shapes and lengths, 1 .. 30 000 of them"""
    rnd = random.Random(seed)
    n = rnd.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 100, 1000, 3000, rnd.randint(1, 30000)])
    shape = rnd.randrange(5)
    names = set()
    while len(names) < n:
        i = len(names)
        if shape == 0:
            nm = "ptg%06dl" % i
        elif shape == 1:
            nm = "h%dtg%06dl" % (1 + i % 2, i)
        elif shape == 2:
            nm = "".join(rnd.choice("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_|.-") for _ in range(rnd.randint(1, 24)))
        elif shape == 3:
            nm = "chr%d_%s" % (i, rnd.choice(["MATERNAL", "PATERNAL"]))
        else:
            nm = str(rnd.randrange(1 << 30))
        names.add(nm)
    names = list(names)
    rnd.shuffle(names)
    lens, sd, te = [], [], []
    for nm in names:
        L = rnd.randint(400, 3000)
        lens.append("%s\t%d\n" % (nm, L))
        if rnd.random() < 0.9:
            a = rnd.randint(0, 50)
            b = rnd.randint(L - 50, L)
            sd.append("%s\t%d\t%d\n" % (nm, a, b))
            s0 = rnd.randint(a + 100, max(a + 100, b - 160)) if b - a > 300 else a
            e0 = min(b, s0 + rnd.choice([24, 30, 48, 60]))
            te.append("%s\t%d\t%d\t%d\t%d\t%d\n" % (nm, L, rnd.randint(0, 1), s0, e0, e0 - s0))
    rnd.shuffle(sd)
    rnd.shuffle(te)
    pl, ps, pt = (os.path.join(tmp, x) for x in ("k.lens", "k.sdust", "k.telomere"))
    open(pl, "w").write("".join(lens))
    open(ps, "w").write("".join(sd))
    open(pt, "w").write("".join(te))
    args = ["telobreaks", pl, ps, pt]
    rr = run(REF, args)
    gg = run(CLI, args)
    if rr[0] in ("timeout", -6, -11):
        return True, None
    ok = (gg[0], gg[1]) == (rr[0], rr[1])
    return ok, None if ok else (n, shape, gg[0], rr[0], len(gg[1]), len(rr[1]), gg[2][-200:])


def fuzz_bigenough(seed, tmp):
    """bigenough (host only on both sides: src/bigenough_main.c:92-296): random assembly bed and boring-bits bed, -T 0 .. 100, the readfish csv (-r), malformed entries"""
    rnd = random.Random(seed)
    n = rnd.randint(1, 40)
    style = rnd.randrange(3)
    names = [("ctg%d" % i, "h%dtg%06dl" % (1 + i % 2, i), "chr%d_%s" % (i // 2 + 1, "MATERNAL" if i % 2 else "PATERNAL"))[style] for i in range(n)]
    lens = [rnd.choice([1, 10, 1000, 50000, 3_000_000, 250_000_000, 2_147_483_647, 3_000_000_000]) if rnd.random() < 0.3 else rnd.randint(1, 5_000_000) for _ in range(n)]
    chroms = "".join("%s\t0\t%d\n" % (nm, L) for nm, L in zip(names, lens))
    rows = []
    for nm, L in zip(names, lens):
        if rnd.random() < 0.25:
            continue
        p = 0
        for _ in range(rnd.randint(1, 6)):
            if p >= L:
                break
            a = rnd.randint(p, min(L - 1, p + max(1, L // 3)))
            b = rnd.randint(a + 1, min(L, a + 1 + max(1, L // 2)))
            rows.append("%s\t%d\t%d\n" % (nm, a, b))
            p = b
    if rnd.random() < 0.3:
        rnd.shuffle(rows)
    kind = rnd.random()
    if kind < 0.05 and rows:
        rows[rnd.randrange(len(rows))] = "nosuchctg\t0\t10\n"
    elif kind < 0.10 and rows:
        rows[rnd.randrange(len(rows))] = "\n"
    elif kind < 0.15 and rows:
        f = rows[0].split("\t")
        rows[0] = "%s\t%s\t%s" % (f[0], f[2].strip(), f[1]) + "\n"          # end < start
    elif kind < 0.20:
        chroms += chroms.splitlines(True)[0]                                  # a contig twice in the assembly bed
    elif kind < 0.25 and rows:
        rows[-1] = rows[-1].rstrip("\n")                                      # no last newline
    elif kind < 0.30 and rows:
        rows[0] = rows[0].replace("\t", " ")
    pc, pb, pr_, pr2 = (os.path.join(tmp, x) for x in ("chroms.bed", "in.bed", "ref.csv", "got.csv"))
    open(pc, "w").write(chroms)
    open(pb, "w").write("".join(rows))
    T = rnd.choice([None, "0", "1", "33", "50", "99", "100", "101", "-1"])
    base = ["bigenough"] + (["-T", T] if T is not None else []) + [pc, pb]
    for f in (pr_, pr2):
        if os.path.exists(f):
            os.remove(f)
    with_csv = rnd.random() < 0.7
    rr = run(REF, base + (["-r", pr_] if with_csv else []))
    gg = run(CLI, base + (["-r", pr2] if with_csv else []))
    csv_r = open(pr_, "rb").read() if os.path.exists(pr_) else None
    csv_g = open(pr2, "rb").read() if os.path.exists(pr2) else None
    ok = (gg[0], gg[1], csv_g) == (rr[0], rr[1], csv_r)
    return ok, None if ok else (base[1:3], n, kind, gg[0], rr[0], len(gg[1]), len(rr[1]), csv_g == csv_r, gg[2][-200:], rr[2][-200:])


def fuzz_panel(seed, tmp):
    rnd = random.Random(seed)
    w = rnd.choice([1, 2, 7, 50, 64, 100, 300, 777, 2500])
    inc = rnd.choice([1, 2, 7, 49, 50, 51, 64, 65, 100, 299, 301, 350, 1000, 2600])
    lens = []
    for c in range(rnd.randint(1, 5)):
        r = rnd.random()
        L = rnd.randint(1, 200) if r < 0.3 else rnd.choice([inc, inc + 1, inc * 3, inc * 3 + 1, inc * 2 + w, w, w + 1, w + 51, inc * 5 + rnd.randint(0, w), 256 * inc + rnd.randint(-1, 1)]) if r < 0.6 else rnd.randint(200, 9000)
        lens.append(max(1, min(L, 30000)))
    t, q = [], []
    hi = rnd.choice([60, 60, 3, 1000, 65535])
    for ci, L in enumerate(lens):
        for p in range(L):
            d = rnd.randint(0, hi)
            t.append("c%d\t%d\t%d\t%d\n" % (ci, p, p + 1, d))
            q.append("c%d\t%d\t%d\t%d\n" % (ci, p, p + 1, rnd.randint(0, d)))
    bad = rnd.random() < 0.15
    if bad and len(t) > 3:
        # one of the reference's five checks (src/boringbits_main.c:204-287): contig names differ, starts differ, a line that is not one position
        k = rnd.randrange(len(t))
        f = t[k].rstrip("\n").split("\t")
        kind = rnd.randrange(4)
        if kind == 0:
            q[k] = "x" + q[k]
        elif kind == 1:
            f[2] = str(int(f[2]) + 1)
            t[k] = "\t".join(f) + "\n"
        elif kind == 2:
            f[1] = str(int(f[1]) + 1)
            f[2] = str(int(f[2]) + 1)
            t[k] = "\t".join(f) + "\n"
        else:
            q = q[:-1]
    # token layouts that fscanf("%s\t%d\t%d\t%d\n") reads the same way or refuses in its own way (src/boringbits_main.c:205-262): any white space between the
    # tokens, a record over two lines, blank lines, signs and leading zeros, depths beyond 65535 (cut, with a warning), a contig whose first line does not
    # start at 0 (not checked), a name that comes back later (a new contig), five columns, a missing column, a token that is no number, no last newline
    odd = rnd.random() < 0.35
    if odd and len(t) > 6:
        for _ in range(rnd.randint(1, 4)):
            k = rnd.randrange(len(t))
            kind = rnd.randrange(12)
            if k >= len(q):
                continue
            ft, fq = t[k].rstrip("\n").split("\t"), q[k].rstrip("\n").split("\t")
            if len(ft) != 4 or len(fq) != 4:             # (a line an earlier turn of this loop has changed)
                continue
            if kind == 0:
                t[k] = " ".join(ft) + "\n"
            elif kind == 1:
                t[k] = ft[0] + "\n" + ft[1] + "\t\t" + ft[2] + " \n\n" + ft[3] + "\n"
            elif kind == 2:
                q[k] = "\n\n" + q[k] + "\n"
            elif kind == 3:
                t[k] = "\t".join([ft[0], "+" + ft[1], "0" + ft[2], "00" + ft[3]]) + "\n"
            elif kind == 4:
                t[k] = "\t".join(ft[:3] + [str(rnd.choice([65535, 65536, 70000, 1000000]))]) + "\n"
            elif kind == 5:
                q[k] = "\t".join(fq[:3] + [str(rnd.choice([65536, 99999]))]) + "\n"
            elif kind == 6:
                t[k] = t[k].rstrip("\n") + "\textra\n"
            elif kind == 7:
                t[k] = "\t".join(ft[:3]) + "\n"
            elif kind == 8:
                q[k] = "\t".join([fq[0], fq[1], fq[2], "x" + fq[3]]) + "\n"
            elif kind == 9:
                t[k] = "\t".join(ft[:3] + ["-" + ft[3]]) + "\n"
            elif kind == 10 and k + 1 < len(t):
                # the rest of this contig under another name, and the old name again later
                name = ft[0]
                j = k
                while j < len(t) and j < len(q) and t[j].startswith(name + "\t"):
                    t[j] = "z" + t[j]
                    q[j] = "z" + q[j]
                    j += 1
            else:
                t[-1] = t[-1].rstrip("\n")
                q[-1] = q[-1].rstrip("\n")
    a, b = os.path.join(tmp, "t.bg"), os.path.join(tmp, "q.bg")
    open(a, "w").write("".join(t))
    open(b, "w").write("".join(q))
    if rnd.random() < 0.15:
        for pth in (a, b):
            data_ = open(pth, "rb").read()
            with gzip.open(pth + ".gz", "wb") as f:
                f.write(data_)
        a, b = a + ".gz", b + ".gz"
    args = [rnd.choice(["noboringbits", "boringbits"]), a, "-q", b, "-w", str(w), "-i", str(inc), "-m", str(rnd.choice([1, 100, 1000, 100000])),
            "-e", str(rnd.choice([0, 5, 100, 5000])), "-L", rnd.choice(["0.4", "0.2", "0.9"]), "-H", rnd.choice(["2.5", "1.2", "3"]), "-Q", rnd.choice(["0.4", "0.9", "0.1"])]
    if rnd.random() < 0.4:
        # the option table as getopt_long reads it (src/boringbits_main.c:45-64, :576-640): long spellings with and without '=', options behind the
        # file, the options the reference takes and ignores (-t -K -B -v -o, --debug-break, --profile-cpu, --accel=yes), values its checks refuse
        longs = {"-w": "--window-size", "-i": "--window-inc", "-m": "--min-ctg-len", "-e": "--edge-len", "-L": "--low-thresh", "-H": "--high-thresh", "-Q": "--low-mq-thresh", "-q": "--qual"}
        cmd, rest = args[0], args[1:]
        pos, opts = rest[0], []
        it = iter(rest[1:])
        for o in it:
            v = next(it)
            k = rnd.random()
            if k < 0.3:
                opts.append([longs[o] + "=" + v])
            elif k < 0.5:
                opts.append([longs[o], v])
            elif k < 0.6:
                opts.append([o + v])
            else:
                opts.append([o, v])
        for extra in rnd.sample([["-t", "4"], ["-K", "100"], ["-B", "1M"], ["-v", "3"], ["--threads", "2"], ["--debug-break=1"], ["--profile-cpu=yes"]] + ([] if os.environ.get("CORNETTO_ACCEL") == "no" else [["--accel=yes"]]) + [["-o", os.path.join(tmp, "ignored.out")],
                                 ["-t", "0"], ["-K", "0"], ["-K", "-3"]], rnd.randint(0, 3)):
            opts.append(extra)
        rnd.shuffle(opts)
        cut = rnd.randint(0, len(opts))
        args = [cmd] + [x for o in opts[:cut] for x in o] + [pos] + [x for o in opts[cut:] for x in o]
    rr = run(REF, args)
    env = rnd.choice([{}, {}, {"CORNETTO_DEVICES": "0,0,0"}, {"CORNETTO_DEVICES": "0,0", "CORNETTO_BG_SHARD_MIN": "1"}, {"CORNETTO_BG_PIECE": "4096"}])
    gg = run(CLI, args, env)
    ok = (gg[0], gg[1]) == (rr[0], rr[1])
    return ok, None if ok else (args, lens, env, bad, odd, gg[0], rr[0], len(gg[1]), len(rr[1]), gg[2][-300:], rr[2][-200:])


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    if not os.path.exists(REF):
        print("oracle/_ref/cornetto is not built (python -c 'import __graft_entry__ as g; g.build()' where /root/reference is)")
        return 2
    bad = 0
    stats = {}
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        for kind, fn in (("fasta", fuzz_fasta), ("telo", fuzz_telo), ("panel", fuzz_panel), ("bigenough", fuzz_bigenough), ("khash", fuzz_khash)):
            if what not in (kind, "all"):
                continue
            nb = 0
            for seed in range(s0, s0 + n):
                ok, info = fn(seed, tmp)
                if not ok:
                    nb += 1
                    print("%s seed %d: MISMATCH %r" % (kind, seed, info), flush=True)
            stats[kind] = (n, nb)
            bad += nb
    print("fuzz_cli: " + ", ".join("%s %d seeds from %d, %d mismatches" % (k, v[0], s0, v[1]) for k, v in stats.items()))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
