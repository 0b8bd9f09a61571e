"""CPU: the host-only parts of the `cornetto` CLI (dispatcher, fa2bed, seq, bigenough, depth stub, usage and
exit codes, the text parsers in front of the device calls) against golden stdout of the unmodified reference.  No GPU
needed: none of these touch HIP.  Every test runs twice: with the product binary and with the sanitized build of the same
host C (`make -C cornetto_amd asan=1`: AddressSanitizer + UndefinedBehaviorSanitizer, the reference's own switch,
Makefile:32-35 / test/test.sh:16-22) — a sanitizer report fails the test."""
import os
import subprocess

import pytest

import cornetto_amd
from helpers import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


ASAN_PATH = cornetto_amd.CLI_PATH + "_asan"


@pytest.fixture(scope="module", params=["product", "asan"])
def cli(request):
    if not os.path.exists(cornetto_amd.CLI_PATH):
        cornetto_amd.build()
    if request.param == "product":
        return cornetto_amd.CLI_PATH
    from helpers import build_asan_cli
    return build_asan_cli()


def run(cli, args, cwd=None, stdin=None):
    # leaks are reported without changing the exit code: a leak on a path that ends in success fails the test; a run that
    # ends in exit(EXIT_FAILURE) leaves with its buffers allocated, as the reference does (src/error.h:97-103)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", LSAN_OPTIONS="exitcode=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([cli] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=cwd, env=env, input=stdin)
    tail = p.stderr.decode(errors="replace")[-3000:]
    assert b"ERROR: AddressSanitizer" not in p.stderr and b"runtime error:" not in p.stderr and p.returncode != 99, tail
    assert p.returncode != 0 or b"LeakSanitizer" not in p.stderr, tail
    return p.returncode, p.stdout, p.stderr


def test_usage_and_version(cli):
    rc, out, err = run(cli, [])
    assert rc == 1 and out == b"" and b"Usage: cornetto <command>" in err
    rc, out, err = run(cli, ["--version"])
    assert rc == 0 and out == b"cornetto 0.2.0\n"
    rc, out, err = run(cli, ["-V"])
    assert rc == 0 and out == b"cornetto 0.2.0\n"
    rc, out, err = run(cli, ["--help"])
    assert rc == 0 and b"Usage: cornetto <command>" in out
    rc, out, err = run(cli, ["nonsense"])
    assert rc == 1 and b"Unrecognised command nonsense" in err


def test_subcommand_usage_exit_codes(cli):
    for sub in ("telofind", "telowin", "sdust", "boringbits", "noboringbits", "bigenough", "fa2bed", "seq", "depth"):
        rc, out, err = run(cli, [sub])
        assert rc == 1, sub
        assert out == b"", sub
    for sub in ("boringbits", "noboringbits", "bigenough", "fa2bed", "seq", "depth"):
        rc, out, err = run(cli, [sub, "-h"])
        assert rc == 0 and b"Usage" in out, sub
    rc, out, err = run(cli, ["noboringbits", "x.bg"])          # -q missing
    assert rc == 1
    rc, out, err = run(cli, ["telowin", "only_one_arg"])
    assert rc == 1


@pytest.mark.parametrize("fa,exp", [("probe.fa", "probe.fa2bed.exp"), ("mix.fa.gz", "mix.fa2bed.exp"), ("reads.fq", "reads.fa2bed.exp")])
def test_fa2bed(cli, golden_dir, fa, exp):
    rc, out, err = run(cli, ["fa2bed", os.path.join(golden_dir, fa)])
    assert rc == 0 and out == golden(golden_dir, exp)
    assert b"Real time:" in err                                 # the 3-line footer of src/main.c:145-149


def test_seq(cli, golden_dir):
    rc, out, err = run(cli, ["seq", "-m", "100", os.path.join(golden_dir, "reads.fq")])
    assert rc == 0 and out == golden(golden_dir, "reads.m100.seq.exp")
    assert b"total reads: 40\t" in err
    rc, out, err = run(cli, ["seq", os.path.join(golden_dir, "reads.fq")])
    assert rc == 0 and out == golden(golden_dir, "reads.default.seq.exp")


@pytest.mark.parametrize("chroms,bed,T,exp_bed,exp_csv", [
    ("chroms.bed", "in.boringbits.bed", None, "out.boringbits.bed", "out.boringbits.csv"),
    ("chroms.bed", "in_dip.boringbits.bed", None, "out_dip.boringbits.bed", "out_dip.boringbits.csv"),
    ("ovf_chroms.bed", "ovf_in.bed", "50", "ovf_T50.bed.exp", "ovf_T50.csv.exp"),
    ("ovf_chroms.bed", "ovf_in.bed", "0", "ovf_T0.bed.exp", "ovf_T0.csv.exp"),
    ("ovf_chroms.bed", "ovf_in.bed", "100", "ovf_T100.bed.exp", "ovf_T100.csv.exp"),
    ("ovf_chroms.bed", "ovf_in.bed", "33", "ovf_T33.bed.exp", "ovf_T33.csv.exp"),
])
def test_bigenough(cli, golden_dir, tmp_path, chroms, bed, T, exp_bed, exp_csv):
    """the reference's own fixture (test/test.sh:33-39) + the int32-overflow cases"""
    d = os.path.join(golden_dir, "bigenough")
    csv = str(tmp_path / "a.csv")
    args = ["bigenough"] + (["-T", T] if T else []) + [os.path.join(d, chroms), os.path.join(d, bed), "-r", csv]
    rc, out, err = run(cli, args)
    assert rc == 0
    assert out == golden(d, exp_bed)
    assert open(csv, "rb").read() == golden(d, exp_csv)
    assert b"Final panel length:" in err


def test_bigenough_errors(cli, golden_dir, tmp_path):
    d = os.path.join(golden_dir, "bigenough")
    bad = tmp_path / "bad.bed"
    bad.write_bytes(b"nosuchctg\t0\t10\n")
    rc, out, err = run(cli, ["bigenough", os.path.join(d, "chroms.bed"), str(bad)])
    assert rc == 1 and b"not found in assembly bed" in err
    bad.write_bytes(b"\n")
    rc, out, err = run(cli, ["bigenough", os.path.join(d, "chroms.bed"), str(bad)])
    assert rc == 1 and b"Malformed bed entry" in err
    dup = tmp_path / "dup.bed"
    dup.write_bytes(b"a\t0\t10\na\t0\t20\n")
    rc, out, err = run(cli, ["bigenough", str(dup), str(bad)])
    assert rc == 1 and b"duplicated" in err
    rc, out, err = run(cli, ["bigenough", "-T", "101", os.path.join(d, "chroms.bed"), str(bad)])
    assert rc == 1


def test_depth_stub(cli):
    rc, out, err = run(cli, ["depth", "whatever.bam"])          # never opened, exactly like the reference
    assert rc == 0 and out == b"" and b"total entries: 0" in err


def test_missing_input_file_exit_codes(cli):
    assert run(cli, ["fa2bed", "/nonexistent.fa"])[0] == 1
    assert run(cli, ["telofind", "/nonexistent.fa"])[0] == 1
    assert run(cli, ["telowin", "/nonexistent.tsv", "99.9"])[0] == 1
    assert run(cli, ["noboringbits", "/nonexistent.bg", "-q", "/nonexistent2.bg"])[0] == 1


def test_text_parsers_in_front_of_the_device_calls(cli, golden_dir, tmp_path):
    """telowin / telobreaks / (no)boringbits parse their text on the host before anything touches the device: without a GPU
    the run ends with the device error (exit 1) — after the parsers have seen the whole input, malformed lines included"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: these paths are covered by tests/test_gpu_cli.py")
    for tsv in ("probe.telomere", "mix.telofind.exp"):
        rc, out, err = run(cli, ["telowin", os.path.join(golden_dir, tsv), "99.9", "0.4"])
        assert rc == 1 and out == b""
    bad = tmp_path / "bad.telomere"
    bad.write_bytes(b"c1\t100\t0\t0\t12\n" + b"x" * 5000 + b"\n\nc2 7\n")
    assert run(cli, ["telowin", str(bad), "99.9"])[0] == 1
    empty = tmp_path / "empty.telomere"
    empty.write_bytes(b"")
    rc, out, err = run(cli, ["telowin", str(empty), "99.9"])
    assert rc == 0 and out == b""                                   # no contig, nothing to scan: src/telomere_windows.c:65-84
    args = [os.path.join(golden_dir, f) for f in ("tb_many.lens", "tb_many.sdust", "tb_many.telomere")]
    assert run(cli, ["telobreaks"] + args)[0] == 1
    rc, out, err = run(cli, ["noboringbits", os.path.join(golden_dir, "cov-total.bg.gz"), "-q", os.path.join(golden_dir, "cov-mq20.bg.gz")])
    assert rc == 1
    for sub in ("sdust", "telofind"):
        assert run(cli, [sub, os.path.join(golden_dir, "mix.fa.gz")])[0] == 1
        assert run(cli, [sub, "-"], stdin=b">a\nACGT\n")[0] == 1
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"@r c\nACGT\n+\nIIII\n@t\nAC\n+\nII\n@cut\nACGT\n+\nII")
    rc, out, err = run(cli, ["seq", "-m", "3", str(fq)])
    assert out == b"@r\tc\nACGT\n+\nIIII\n"
    fa = tmp_path / "r.fa"
    fa.write_bytes(b">a b c\r\nAC\r\nGT\r\n>e\n\n>f\nA")
    rc, out, err = run(cli, ["fa2bed", str(fa)])
    assert rc == 0 and out == b"a\t0\t4\ne\t0\t0\nf\t0\t1\n"
    # "-" is the standard input for sdust alone (src/sdust/sdust.c:194); seq, fa2bed and telofind hand it to gzopen() as a file name
    # (src/seq.c:106, src/assbed.c:92, src/find_telomere.c:96): F_CHK's message and exit status 1, nothing on stdout
    for sub in (["seq", "-m", "3"], ["fa2bed"]):
        rc, out, err = run(cli, sub + ["-"], stdin=fq.read_bytes())
        assert rc == 1 and out == b"" and b"Could not to open file -: No such file or directory" in err


def _bedgraph_pair(tmp_path, lens, seed, digits_t=(10_000, 60_000), digits_q=(0, 10)):
    import numpy as np
    rng = np.random.default_rng(seed)
    rows_t, rows_q, first = [], [], []
    nl = 0
    for ci, n in enumerate(lens):
        name = "ptg%06dl" % ci if ci % 3 else "c%d" % ci
        d = rng.integers(*digits_t, size=n)
        q = rng.integers(*digits_q, size=n)
        first.append(nl)
        nl += n
        rows_t.append("".join("%s\t%d\t%d\t%d\n" % (name, p, p + 1, v) for p, v in enumerate(d)))
        rows_q.append("".join("%s\t%d\t%d\t%d\n" % (name, p, p + 1, v) for p, v in enumerate(q)))
    a, b = tmp_path / "t.bg", tmp_path / "q.bg"
    a.write_text("".join(rows_t))
    b.write_text("".join(rows_q))
    return str(a), str(b), first


@pytest.mark.parametrize("lens,devices", [
    ([30_000, 5, 20_000, 70_000, 1, 1, 40_000, 9_000], "0,1"),
    ([30_000, 5, 20_000, 70_000, 1, 1, 40_000, 9_000], "0,1,2,3"),
    ([3_000] * 40, "0,1,2,3,4,5,6,7"),
    ([200_000], "0,1,2"),                              # one contig: nothing to cut
    ([100_000, 100_000], "0,1,2,3,4,5,6,7"),           # more devices than contigs
])
def test_bedgraph_cuts_for_sharded_ingest(cli, tmp_path, lens, devices):
    """CORNETTO_DEVICES + two regular files: every device parses its own share of the text.  The cuts (found with a few pread() probes and
    a binary search per cut, before any device is opened) are line starts where the contig changes, and the SAME line in both files
    although their bytes per line differ (5-digit depths against 1-digit ones)"""
    a, b, first = _bedgraph_pair(tmp_path, lens, 5)
    env = dict(os.environ, CORNETTO_DEVICES=devices, CORNETTO_BG_SHARD_MIN="1", CORNETTO_BG_SPLIT_ONLY="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99")
    p = subprocess.run([cli, "noboringbits", a, "-q", b], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0 and b"ERROR: AddressSanitizer" not in p.stderr and b"runtime error:" not in p.stderr, p.stderr.decode(errors="replace")[-2000:]
    cuts = [tuple(int(x) for x in l.split()) for l in p.stdout.decode().splitlines()]
    ta, tb = open(a, "rb").read(), open(b, "rb").read()
    assert len(cuts) <= len(devices.split(",")) - 1
    if len(lens) == 1:
        assert cuts == []
    elif sum(lens) > 100_000 and len(lens) > 2:
        assert len(cuts) >= 1
    last = (0, 0)
    for ct, cq in cuts:
        assert last[0] < ct < len(ta) and last[1] < cq < len(tb)
        assert ta[ct - 1:ct] == b"\n" and tb[cq - 1:cq] == b"\n"
        nl = ta[:ct].count(b"\n")
        assert nl == tb[:cq].count(b"\n") and nl in first          # the same line, and the first of a contig
        last = (ct, cq)
