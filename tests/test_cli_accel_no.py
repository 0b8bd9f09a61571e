"""CPU (no GPU anywhere): the product's own host path — `--accel=no` (the reference's switch, src/boringbits_main.c:627-632) /
CORNETTO_ACCEL=no — through the C CLI, the plain build and the AddressSanitizer + UBSan build, against the SAME golden stdout
of the unmodified reference that the device path is held to (tests/test_gpu_cli.py).  This is BASELINE.json's configuration 1
("CPU only ... plumbing, no GPU") as written.  The host path is cornetto_amd/cli/host_backend.c: product code, it never touches
oracle/ — and it is never chosen silently: without the switch a box without a GPU still exits 1."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import cornetto_amd
from helpers import FASTA_SIDE, PANEL, PANEL_ABORT, golden, panel_argv
from test_oracle_golden import TELOBREAKS_CASES

ASAN_PATH = cornetto_amd.CLI_PATH + "_asan"


@pytest.fixture(scope="module", params=["product", "asan"])
def cli(request):
    if request.param == "product":
        assert os.path.exists(cornetto_amd.CLI_PATH), "build the CLI first (make -C cornetto_amd)"
        return cornetto_amd.CLI_PATH
    from helpers import build_asan_cli
    return build_asan_cli()


def run(cli, args, env=None, stdin=None):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", LSAN_OPTIONS="exitcode=0", UBSAN_OPTIONS="print_stacktrace=1")
    e.pop("CORNETTO_ACCEL", None)
    e.update(env or {})
    p = subprocess.run([cli] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, stdin=stdin)
    assert p.returncode != 99 and b"runtime error" not in p.stderr, p.stderr.decode(errors="replace")[-3000:]
    return p.returncode, p.stdout, p.stderr


HOST = {"CORNETTO_ACCEL": "no", "HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""}


@pytest.fixture(scope="module")
def plain(golden_dir, tmp_path_factory):
    d = tmp_path_factory.mktemp("plain_host")
    out = {}
    for fn in ("cov-total.bg.gz", "cov-mq20.bg.gz", "sparse-total.bg.gz", "sparse-mq20.bg.gz"):
        dst = d / fn[:-3]
        dst.write_bytes(gzip.open(os.path.join(golden_dir, fn)).read())
        out[fn[:-3]] = str(dst)
    return out


@pytest.mark.parametrize("args,exp", FASTA_SIDE)
def test_fasta_side_on_the_host(cli, golden_dir, args, exp):
    if cli == ASAN_PATH and exp == "mix.w100t25.sdust.exp":
        pytest.skip("a minute under the sanitizers (find_perfect over a 98-word window at every base of the repeats); the plain build runs it")
    a = [os.path.join(golden_dir, x) if os.path.exists(os.path.join(golden_dir, x)) else x for x in args]
    rc, out, err = run(cli, a, HOST)
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, exp)


def test_sdust_stdin_on_the_host(cli, golden_dir):
    with open(os.path.join(golden_dir, "probe_sdust.fa"), "rb") as f:
        rc, out, _ = run(cli, ["sdust", "-"], HOST, stdin=f)
    assert rc == 0 and out == golden(golden_dir, "probe_sdust.sdust.exp")


@pytest.mark.parametrize("how", ["--accel=no", "--accel no", "env"])
@pytest.mark.parametrize("args,exp", PANEL)
def test_panel_on_the_host(cli, golden_dir, plain, args, exp, how):
    """the option sets of the reference's own test/test.sh:25,29, defaults and odd window sizes; --accel=no is the reference's
    own spelling of the choice (getopt_long: `--accel=no` and `--accel no` alike), CORNETTO_ACCEL=no the environment's"""
    a = panel_argv(plain, args)
    env = {"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""}
    if how == "env":
        env = HOST
    else:
        a = a[:1] + how.split() + a[1:]
    rc, out, err = run(cli, a, env)
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, exp)
    assert err.count(b"Average depth:") == 1


@pytest.mark.parametrize("args,exp", PANEL_ABORT)
def test_panel_dies_where_the_reference_asserts_on_the_host(cli, golden_dir, plain, args, exp):
    """get_regs() runs over EVERY contig (it knows no -m) before anything is printed: assert(st<end), src/boringbits_main.c:353 -> SIGABRT,
    empty stdout.  (The sanitizer build reports the abort as well; its status is the signal's either way.)"""
    a = panel_argv(plain, args)
    rc, out, err = run(cli, a[:1] + ["--accel=no"] + a[1:], {"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert rc == -6, (rc, err.decode())
    assert out == golden(golden_dir, exp) == b""
    assert b"src/boringbits_main.c:353: get_regs: Assertion `st<end' failed." in err


@pytest.mark.parametrize("lens_f,sd_f,tel_f,exp", TELOBREAKS_CASES)
def test_telobreaks_on_the_host(cli, golden_dir, lens_f, sd_f, tel_f, exp):
    rc, out, err = run(cli, ["telobreaks"] + [os.path.join(golden_dir, f) for f in (lens_f, sd_f, tel_f)], HOST)
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, exp)


def test_malformed_bedgraphs_exit_1_on_the_host(cli, plain, tmp_path):
    """the reference's five checks (src/boringbits_main.c:209-227,249-259), in its order, with its messages"""
    tot = open(plain["cov-total.bg"], "rb").read().splitlines(True)
    mq = open(plain["cov-mq20.bg"], "rb").read().splitlines(True)

    def attempt(t, q):
        a, b = tmp_path / "t.bg", tmp_path / "q.bg"
        a.write_bytes(b"".join(t))
        b.write_bytes(b"".join(q))
        rc, out, err = run(cli, ["noboringbits", "--accel=no", str(a), "-q", str(b)])
        return rc, err

    assert attempt(tot[:50], mq[:50])[0] == 0
    rc, err = attempt([b"track type=bedGraph\n"] + tot[:50], [b"track type=bedGraph\n"] + mq[:50])
    assert rc == 1 and b"should have 4 columns. Had 1." in err
    rc, err = attempt(tot[:50], mq[:49])
    assert rc == 1 and b"not in the same order" in err
    rc, err = attempt(tot[:50], mq[:40] + mq[41:51])
    assert rc == 1 and b"not in the same order" in err
    rc, err = attempt(tot[:20] + tot[21:50], mq[:20] + mq[21:50])
    assert rc == 1 and b"incremantal at one base resolution. Found 19 to 21" in err
    rl = [b"ptg000001l\t0\t5\t30\n"]
    rc, err = attempt(rl, rl)
    assert rc == 1 and b"end=start+1. Found 0 to 5" in err
    rc, err = attempt([b"c\t0\t1\t70000\nc\t1\t2\t-3\n"], [b"c 0 1 5 c 1\n2\n7"])     # tokens, not lines; clamp; a negative value wraps
    assert rc == 0 and b"truncated to 65535" in err
    rc, err = attempt([b"c\t0\t1\n"], [b"c\t0\t1\t3\n"])
    assert rc == 1 and b"Had 3." in err
    assert attempt([], [])[0] == 0                                                   # no record at all: nothing printed, exit 0


def test_host_path_against_random_sequences_and_the_oracle(cli, tmp_path):
    """beyond the goldens: random records with repeats, N runs, lower case and other bytes — the host path and the oracle agree
    (the oracle is the checker here, never part of the product)"""
    import oracle_bind as ob
    rng = np.random.default_rng(11)
    recs = []
    for i in range(30):
        n = int(rng.integers(0, 4000))
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
        for _ in range(int(rng.integers(0, 6))):
            if n < 50:
                break
            p, l = int(rng.integers(0, n - 40)), int(rng.integers(3, 300))
            unit = [b"TTAGGG", b"CCCTAA", b"A", b"AC", b"N", b"acg", b"CATTC", b"R"][int(rng.integers(0, 8))]
            rep = np.frombuffer((unit * (l // len(unit) + 1))[:l], dtype=np.uint8)
            s[p:p + l] = rep[:len(s[p:p + l])]
        recs.append((b"r%d" % i, s))
    fa = tmp_path / "r.fa"
    fa.write_bytes(b"".join(b">" + n + b" x\n" + s.tobytes() + b"\n" for n, s in recs))
    for T, W in ((20, 64), (10, 32), (25, 100), (5, 7), (2, 3)):
        rc, out, err = run(cli, ["sdust", "-w", str(W), "-t", str(T), str(fa)], HOST)
        assert rc == 0, err.decode()
        exp = b"".join(b"%s\t%d\t%d\n" % (n, int(x) >> 32, int(x) & 0xFFFFFFFF) for n, s in recs for x in ob.sdust(s, T, W))
        assert out == exp, (T, W)
    for motif in (b"TTAGGG", b"AAAA", b"ACA", b"CATTCCATTC"):
        rc, out, err = run(cli, ["telofind", str(fa), motif.decode()], HOST)
        assert rc == 0, err.decode()
        exp = b"".join(b"%s\t%d\t%d\t%d\t%d\t%d\n" % (n, len(s), h["strand"], h["start"], h["end"], h["end"] - h["start"])
                       for n, s in recs for h in ob.telofind(s, motif))
        assert out == exp, motif


def test_the_host_path_is_never_chosen_silently(cli, golden_dir):
    """without the switch a process that finds no GPU exits 1 with the reason (no silent CPU fallback of the product path);
    sub-commands that never needed a device are not affected by the switch"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the device path would simply run")
    rc, out, err = run(cli, ["sdust", os.path.join(golden_dir, "probe_sdust.fa")])
    assert rc == 1 and out == b"" and b"cannot open HIP device" in err
    rc, out, err = run(cli, ["fa2bed", os.path.join(golden_dir, "probe.fa")], HOST)
    assert rc == 0 and out == golden(golden_dir, "probe.fa2bed.exp")
