"""GPU: the `cornetto` CLI end to end (host C -> C ABI -> HIP kernels) against golden stdout of the
unmodified reference, byte for byte — the drop-in claim of the panel path."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import cornetto_amd
from helpers import FASTA_SIDE, PANEL, PANEL_ABORT, golden, panel_argv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cli():
    assert os.path.exists(cornetto_amd.CLI_PATH), "build the CLI first (make -C cornetto_amd)"
    return cornetto_amd.CLI_PATH


def run(cli, args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([cli] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    return p.returncode, p.stdout, p.stderr


@pytest.fixture(scope="module")
def plain(golden_dir, tmp_path_factory):
    """uncompressed copies of the gz fixtures (bedgraphs are read with fopen by the reference)"""
    d = tmp_path_factory.mktemp("plain")
    out = {}
    for fn in ("cov-total.bg.gz", "cov-mq20.bg.gz", "sparse-total.bg.gz", "sparse-mq20.bg.gz", "mix.fa.gz"):
        dst = d / fn[:-3]
        dst.write_bytes(gzip.open(os.path.join(golden_dir, fn)).read())
        out[fn[:-3]] = str(dst)
    return out


@pytest.mark.parametrize("args,exp", FASTA_SIDE)
def test_fasta_side(cli, golden_dir, args, exp):
    a = [os.path.join(golden_dir, x) if os.path.exists(os.path.join(golden_dir, x)) else x for x in args]
    rc, out, err = run(cli, a)
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, exp)


def test_batched_reads_same_output(cli, golden_dir):
    """tiny batches (one device pass per ~500 bases) must not change the output"""
    for sub, exp in (("sdust", "reads.sdust.exp"), ("telofind", None)):
        rc1, out1, _ = run(cli, [sub, os.path.join(golden_dir, "reads.fq")])
        rc2, out2, _ = run(cli, [sub, os.path.join(golden_dir, "reads.fq")], env={"CORNETTO_BATCH_BASES": "500"})
        assert rc1 == 0 and rc2 == 0 and out1 == out2
        if exp:
            assert out1 == golden(golden_dir, exp)


def test_sdust_stdin(cli, golden_dir):
    data = open(os.path.join(golden_dir, "probe_sdust.fa"), "rb").read()
    p = subprocess.run([cli, "sdust", "-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == golden(golden_dir, "probe_sdust.sdust.exp")




@pytest.mark.parametrize("args,exp", PANEL)
def test_panel(cli, golden_dir, plain, args, exp):
    """the two option sets of the reference's own test/test.sh:25,29 plus defaults and odd window sizes"""
    a = panel_argv(plain, args)
    rc, out, err = run(cli, a)
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, exp)


@pytest.mark.parametrize("env", [{}, {"CORNETTO_DEVICES": "0,0,0"}, {"CORNETTO_DEVICES": "0,0", "CORNETTO_BG_SHARD_MIN": "1"}, {"CORNETTO_BG_PIECE": "4096"}])
@pytest.mark.parametrize("args,exp", PANEL_ABORT)
def test_panel_dies_where_the_reference_asserts(cli, golden_dir, plain, args, exp, env):
    """-i larger than -w with a contig whose last window would be empty: get_regs() runs over every contig before the reference prints
    anything and its assert(st<end) (src/boringbits_main.c:353) raises SIGABRT — same status, nothing on stdout, on one device, with the
    contigs dealt to several, and with the text itself cut into shares"""
    rc, out, err = run(cli, panel_argv(plain, args), env)
    assert rc == -6, (rc, err.decode())
    assert out == golden(golden_dir, exp) == b""
    assert b"src/boringbits_main.c:353: get_regs: Assertion `st<end' failed." in err


def test_panel_random_windows_against_the_reference_binary(cli, tmp_path):
    """(no)boringbits on random per-base bedgraph pairs with random -w / -i (smaller, equal, larger than each other), -m, -e and thresholds, contig
    lengths at and around multiples of -i: stdout AND exit status of the device path against the unmodified reference's (oracle/_ref/cornetto,
    which travels with the snapshot) — including the command lines on which the reference dies of its assert (SIGABRT, empty stdout)"""
    import random
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "cornetto")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/cornetto not built")
    rnd = random.Random(2025)
    aborts = ok = 0
    for it in range(48):
        w = rnd.choice([1, 2, 7, 50, 64, 100, 300, 777, 2500])
        inc = rnd.choice([1, 2, 7, 49, 50, 51, 64, 65, 100, 299, 301, 350, 1000, 2600])
        lens = []
        for c in range(rnd.randint(1, 4)):
            r = rnd.random()
            L = rnd.randint(1, 200) if r < 0.3 else rnd.choice([inc, inc + 1, inc * 3, inc * 3 + 1, inc * 2 + w, w, w + 1, w + 51, inc * 5 + rnd.randint(0, w)]) if r < 0.6 else rnd.randint(200, 6000)
            lens.append(max(1, L))
        t, q = [], []
        for ci, L in enumerate(lens):
            for p in range(L):
                d = rnd.randint(0, 60)
                t.append("c%d\t%d\t%d\t%d\n" % (ci, p, p + 1, d))
                q.append("c%d\t%d\t%d\t%d\n" % (ci, p, p + 1, rnd.randint(0, d)))
        a, b = tmp_path / "t.bg", tmp_path / "q.bg"
        a.write_text("".join(t))
        b.write_text("".join(q))
        args = [rnd.choice(["noboringbits", "boringbits"]), str(a), "-q", str(b), "-w", str(w), "-i", str(inc), "-m", str(rnd.choice([1, 100, 1000, 100000])),
                "-e", str(rnd.choice([0, 5, 100, 5000])), "-L", rnd.choice(["0.4", "0.2", "0.9"]), "-H", rnd.choice(["2.5", "1.2", "3"]), "-Q", rnd.choice(["0.4", "0.9", "0.1"])]
        pr = subprocess.run([ref] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        env = {"CORNETTO_DEVICES": "0,0", "CORNETTO_BG_SHARD_MIN": "1"} if it % 4 == 3 else {"CORNETTO_DEVICES": "0,0,0"} if it % 4 == 2 else {}
        rc, out, err = run(cli, args, env)
        assert (rc, out) == (pr.returncode, pr.stdout), (args, lens, env, rc, pr.returncode, err.decode()[-300:])
        aborts += pr.returncode == -6
        ok += pr.returncode == 0
    assert aborts >= 8 and ok >= 20


def test_panel_format_error_wins_over_the_assert(cli, plain, tmp_path):
    """the reference parses both files completely (exit 1 on a malformed line) before get_regs() can assert"""
    tot = open(plain["cov-total.bg"], "rb").read().splitlines(True)
    mq = open(plain["cov-mq20.bg"], "rb").read().splitlines(True)
    a, b = tmp_path / "t.bg", tmp_path / "q.bg"
    a.write_bytes(b"".join(tot[:-1]) + b"ptg000007l\t9999\t10001\t3\n")
    b.write_bytes(b"".join(mq[:-1]) + b"ptg000007l\t9999\t10001\t3\n")
    for env in ({}, {"CORNETTO_DEVICES": "0,0", "CORNETTO_BG_SHARD_MIN": "1"}):
        rc, out, err = run(cli, ["noboringbits", str(a), "-q", str(b), "-w", "300", "-i", "350"], env)
        assert rc == 1 and out == b"", (rc, err.decode())


def test_panel_accel_yes_and_the_ignored_options(cli, golden_dir, plain):
    """--accel=yes is the reference's spelling of the default here; -t -K -B -o --debug-break --profile-cpu are accepted and ignored
    (src/boringbits_main.c:590-632)"""
    a = panel_argv(plain, ["noboringbits", "T", "-q", "Q", "-t", "4", "-K", "10", "-B", "1M", "-o", "/dev/null", "--debug-break", "1", "--profile-cpu", "yes", "--accel=yes"])
    rc, out, err = run(cli, a)
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, "bg.fun_default.exp")


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0,0", "0,0,0,0,0,0,0,0,0,0,0,0"])
@pytest.mark.parametrize("args,exp", PANEL)
def test_panel_window_stage_over_several_devices(cli, golden_dir, plain, args, exp, devices):
    """CORNETTO_DEVICES: the contigs are dealt to the devices after the ingest (cornetto_cov_shard), every device sums and
    classifies its share, the host adds the three totals up for the common thresholds — same bytes as one device (here the
    same GPU several times; more devices than contigs leaves some without work)"""
    a = panel_argv(plain, args)
    rc, out, err = run(cli, a, {"CORNETTO_DEVICES": devices})
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, exp)
    assert err.count(b"Average depth:") == 1


def test_panel_malformed_inputs_exit_1(cli, plain, tmp_path):
    tot = open(plain["cov-total.bg"], "rb").read().splitlines(True)
    mq = open(plain["cov-mq20.bg"], "rb").read().splitlines(True)

    def attempt(t, q):
        a, b = tmp_path / "t.bg", tmp_path / "q.bg"
        a.write_bytes(b"".join(t))
        b.write_bytes(b"".join(q))
        return run(cli, ["noboringbits", str(a), "-q", str(b)])

    assert attempt(tot[:50], mq[:50])[0] == 0
    assert attempt([b"track type=bedGraph\n"] + tot[:50], [b"track type=bedGraph\n"] + mq[:50])[0] == 1   # 4 columns
    assert attempt(tot[:50], mq[:49])[0] == 1                      # second file shorter
    assert attempt(tot[:50], mq[:40] + mq[41:51])[0] == 1          # not in the same order
    assert attempt(tot[:20] + tot[21:50], mq[:20] + mq[21:50])[0] == 1   # not incremental
    rl = [b"ptg000001l\t0\t5\t30\n"]
    assert attempt(rl, rl)[0] == 1                                 # run-length line: end != start+1


@pytest.mark.parametrize("piece", ["97", "5000", "1000003"])
def test_panel_streamed_in_small_pieces(cli, golden_dir, plain, piece):
    """the CLI streams both bedgraphs to the device parser; piece boundaries anywhere must not matter"""
    args = ["noboringbits", "-H", "2.5", "-L", "0.5", "-Q", "0.5", plain["cov-total.bg"], "-q", plain["cov-mq20.bg"], "-m", "10000", "-e", "1000"]
    rc, out, err = run(cli, args, env={"CORNETTO_BG_PIECE": piece})
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, "bg.fun_t2.exp")


@pytest.mark.parametrize("piece", ["97", "5000"])
def test_panel_sequential_reader_and_fifos(cli, golden_dir, plain, piece, tmp_path):
    """the fread() loop that anything but two regular files takes (CORNETTO_BG_THREADS=0 asks for it by hand; FIFOs get it by themselves)"""
    import threading
    args = ["noboringbits", "-H", "2.5", "-L", "0.5", "-Q", "0.5", plain["cov-total.bg"], "-q", plain["cov-mq20.bg"], "-m", "10000", "-e", "1000"]
    rc, out, err = run(cli, args, env={"CORNETTO_BG_PIECE": piece, "CORNETTO_BG_THREADS": "0"})
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, "bg.fun_t2.exp")
    fifos = [str(tmp_path / "t.fifo"), str(tmp_path / "q.fifo")]
    for f in fifos:
        os.mkfifo(f)

    def pump(src, dst):
        with open(dst, "wb") as o:
            o.write(open(src, "rb").read())
    th = [threading.Thread(target=pump, args=(plain[k], f)) for k, f in zip(("cov-total.bg", "cov-mq20.bg"), fifos)]
    for t in th:
        t.start()
    args[7], args[9] = fifos
    rc, out, err = run(cli, args, env={"CORNETTO_BG_PIECE": piece})
    for t in th:
        t.join()
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, "bg.fun_t2.exp")


@pytest.mark.parametrize("piece", ["4099", "70001", "1048576"])
def test_panel_threaded_reader_with_unequal_line_lengths(cli, piece, tmp_path):
    """two regular files are read by pread() threads one round ahead of the parser; the files spend different numbers of bytes per
    line (5-digit depths against 1-digit ones), so equal byte counts per round would let one file run ahead without bound: the
    reader sizes the rounds by what is pending.  Same bytes on stdout as the sequential loop."""
    rng = np.random.default_rng(int(piece))
    lens = [40_000, 1, 25_000, 130_000, 7]
    rows_t, rows_q = [], []
    for ci, n in enumerate(lens):
        d = rng.integers(10_000, 60_000, size=n)
        d[rng.integers(0, n, size=n // 50 + 1)] = 3
        q = rng.integers(0, 10, size=n)
        name = "ctg%d_a_rather_long_contig_name" % ci
        pos = np.arange(n)
        rows_t.append("".join("%s\t%d\t%d\t%d\n" % (name, p, p + 1, v) for p, v in zip(pos, d)))
        rows_q.append("".join("%s\t%d\t%d\t%d\n" % (name, p, p + 1, v) for p, v in zip(pos, q)))
    a, b = tmp_path / "t.bg", tmp_path / "q.bg"
    a.write_text("".join(rows_t))
    b.write_text("".join(rows_q))
    args = ["noboringbits", str(a), "-q", str(b), "-m", "20000", "-e", "1000", "-w", "500", "-i", "50"]
    rc0, out0, err0 = run(cli, args, env={"CORNETTO_BG_THREADS": "0"})
    assert rc0 == 0, err0.decode()
    for threads in ("1", "3", "8"):
        rc, out, err = run(cli, args, env={"CORNETTO_BG_PIECE": piece, "CORNETTO_BG_THREADS": threads})
        assert rc == 0, err.decode()
        assert out == out0 and len(out0) > 1000
        assert [l for l in err.splitlines() if l.startswith(b"Average")] == [l for l in err0.splitlines() if l.startswith(b"Average")]


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0,0"])
@pytest.mark.parametrize("args,exp", PANEL)
def test_panel_sharded_ingest_over_several_devices(cli, golden_dir, plain, args, exp, devices):
    """CORNETTO_DEVICES with two regular files: the text itself is cut into shares of whole contigs (the same line in both files), every
    device parses, sums and classifies its share; the host adds the three totals up — same bytes as one device.  (CORNETTO_BG_SHARD_MIN=1:
    the fixtures are far below the 64 MB per device at which the CLI cuts by itself.)"""
    a = panel_argv(plain, args)
    rc, out, err = run(cli, a, {"CORNETTO_DEVICES": devices, "CORNETTO_BG_SHARD_MIN": "1"})
    assert rc == 0, err.decode()
    assert out == golden(golden_dir, exp)
    assert err.count(b"Average depth:") == 1
    rc1, out1, err1 = run(cli, a)
    assert [l for l in err.splitlines() if l.startswith((b"Average", b"Number of contigs"))] == [l for l in err1.splitlines() if l.startswith((b"Average", b"Number of contigs"))]


def test_panel_sharded_ingest_unequal_line_lengths_and_errors(cli, tmp_path):
    """shares of a pair whose files spend different numbers of bytes per line; a malformed line in a LATER share is reported like the
    sequential parse reports it (exit 1), and an error in the first share wins over one in the last"""
    rng = np.random.default_rng(9)
    lens = [40_000, 1, 25_000, 130_000, 7, 60_000, 30_000, 2_600]
    rows_t, rows_q = [], []
    for ci, n in enumerate(lens):
        d = rng.integers(10_000, 60_000, size=n)
        d[rng.integers(0, n, size=n // 50 + 1)] = 3
        q = rng.integers(0, 10, size=n)
        name = "ctg%d_a_rather_long_contig_name" % ci
        rows_t.append(["%s\t%d\t%d\t%d\n" % (name, p, p + 1, v) for p, v in enumerate(d)])
        rows_q.append(["%s\t%d\t%d\t%d\n" % (name, p, p + 1, v) for p, v in enumerate(q)])
    a, b = tmp_path / "t.bg", tmp_path / "q.bg"

    def write(rt, rq):
        a.write_text("".join("".join(r) for r in rt))
        b.write_text("".join("".join(r) for r in rq))
    write(rows_t, rows_q)
    args = ["noboringbits", str(a), "-q", str(b), "-m", "20000", "-e", "1000", "-w", "500", "-i", "50"]
    rc0, out0, err0 = run(cli, args)
    assert rc0 == 0 and len(out0) > 1000, err0.decode()
    for devices in ("0,0", "0,0,0", "0,0,0,0,0,0,0,0"):
        rc, out, err = run(cli, args, {"CORNETTO_DEVICES": devices, "CORNETTO_BG_SHARD_MIN": "1", "CORNETTO_BG_PIECE": "300000"})
        assert rc == 0, err.decode()
        assert out == out0
        assert b"sharded ingest" in err
        assert [l for l in err.splitlines() if l.startswith(b"Average")] == [l for l in err0.splitlines() if l.startswith(b"Average")]
    # a position that repeats in the last contig but one; then also a 3-column line in the first contig
    bad_t = [list(r) for r in rows_t]
    bad_t[6][100] = bad_t[6][99]
    bad_q = [list(r) for r in rows_q]
    bad_q[6][100] = bad_q[6][99]
    write(bad_t, bad_q)
    rc1, out1, err1 = run(cli, args)
    rc, out, err = run(cli, args, {"CORNETTO_DEVICES": "0,0,0", "CORNETTO_BG_SHARD_MIN": "1"})
    assert rc1 == 1 and rc == 1 and b"incremantal" in err and b"incremantal" in err1
    bad_t[0][50] = "ctg0_a_rather_long_contig_name\t50\t51\n"
    write(bad_t, bad_q)
    rc1, out1, err1 = run(cli, args)
    rc, out, err = run(cli, args, {"CORNETTO_DEVICES": "0,0,0", "CORNETTO_BG_SHARD_MIN": "1"})
    assert rc1 == 1 and rc == 1 and b"4 columns" in err1 and b"4 columns" in err
    # one contig's block twice in cov-mq (the same last line in front of every cut): the sequential loop pairs the surplus lines with
    # the next contig's cov-total records and stops with "not in the same order"; a share must not drop them silently
    dup_q = [list(r) for r in rows_q]
    dup_q[1] = dup_q[1] + dup_q[1]
    write(rows_t, dup_q)
    rc1, out1, err1 = run(cli, args)
    rc, out, err = run(cli, args, {"CORNETTO_DEVICES": "0,0,0,0,0", "CORNETTO_BG_SHARD_MIN": "1"})
    assert rc1 == 1 and rc == 1 and b"not in the same order" in err1 and b"not in the same order" in err


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
@pytest.mark.parametrize("batch", ["", "200000"])
@pytest.mark.parametrize("args,exp", [
    (["telofind", "mix.fa.gz"], "mix.telofind.exp"),
    (["telofind", "mix.fa.gz", "AAAA"], "mix.AAAA.telofind.exp"),
    (["sdust", "mix.fa.gz"], "mix.sdust.exp"),
    (["sdust", "-w", "32", "-t", "10", "mix.fa.gz"], "mix.w32t10.sdust.exp"),
    (["sdust", "reads.fq"], "reads.sdust.exp"),
    (["telofind", "probe.fa"], "probe.telofind.exp"),
])
def test_several_devices_print_the_output_of_one(cli, golden_dir, args, exp, devices, batch):
    """CORNETTO_DEVICES: the records of a batch are dealt to one host thread + handle per listed device (LPT by length) and
    printed in input order after the join — here the same GPU several times, which exercises everything but the peer GPUs;
    small batches make several rounds of it"""
    a = [os.path.join(golden_dir, x) if os.path.exists(os.path.join(golden_dir, x)) else x for x in args]
    env = {"CORNETTO_DEVICES": devices}
    if batch:
        env["CORNETTO_BATCH_BASES"] = batch
    rc, out, err = run(cli, a, env)
    assert rc == 0, err[-500:]
    assert out == golden(golden_dir, exp)


def _device_count():
    import cornetto_amd
    return int(cornetto_amd.lib().cornetto_accel_device_count())


@pytest.mark.parametrize("args,exp", [
    (["sdust", "mix.fa.gz"], "mix.sdust.exp"),
    (["telofind", "mix.fa.gz"], "mix.telofind.exp"),
    (["sdust", "reads.fq"], "reads.sdust.exp"),
])
def test_distinct_devices_print_the_output_of_one(cli, golden_dir, args, exp):
    """CORNETTO_DEVICES over DISTINCT GPUs (handles, streams and result buffers of different devices side by side): runs wherever
    at least two are visible, skipped on a one-GPU box"""
    n = _device_count()
    if n < 2:
        pytest.skip("one GPU visible")
    a = [os.path.join(golden_dir, x) if os.path.exists(os.path.join(golden_dir, x)) else x for x in args]
    for devices in ("0,1", ",".join(str(i) for i in range(min(n, 8)))):
        rc, out, err = run(cli, a, {"CORNETTO_DEVICES": devices, "CORNETTO_BATCH_BASES": "200000"})
        assert rc == 0, err[-500:]
        assert out == golden(golden_dir, exp)


@pytest.mark.parametrize("args,exp", PANEL[:3])
def test_panel_window_stage_over_distinct_devices(cli, golden_dir, plain, args, exp):
    """cornetto_cov_shard with peer copies between DISTINCT GPUs (skipped on a one-GPU box)"""
    n = _device_count()
    if n < 2:
        pytest.skip("one GPU visible")
    a = panel_argv(plain, args)
    for devices in ("0,1", "1,0", ",".join(str(i) for i in range(min(n, 8)))):
        rc, out, err = run(cli, a, {"CORNETTO_DEVICES": devices})
        assert rc == 0, err.decode()
        assert out == golden(golden_dir, exp)
        rc, out, err = run(cli, a, {"CORNETTO_DEVICES": devices, "CORNETTO_BG_SHARD_MIN": "1"})       # and the sharded ingest
        assert rc == 0, err.decode()
        assert out == golden(golden_dir, exp)


def test_device_list_errors(cli, golden_dir):
    rc, out, err = run(cli, ["sdust", os.path.join(golden_dir, "probe.fa")], {"CORNETTO_DEVICES": "0,banana"})
    assert rc == 1 and out == b"" and b"CORNETTO_DEVICES" in err
    rc, out, err = run(cli, ["sdust", os.path.join(golden_dir, "probe.fa")], {"CORNETTO_DEVICES": "0,63"})
    assert rc == 1 and out == b"" and b"cannot open HIP device 63" in err
    rc, out, err = run(cli, ["sdust", os.path.join(golden_dir, "probe.fa")], {"CORNETTO_DEVICES": "0"})       # one device: the usual path
    assert rc == 0 and out == golden(golden_dir, "probe.sdust.exp")


@pytest.mark.parametrize("env", [{}, {"CORNETTO_DEVICES": "0,0,0"}, {"CORNETTO_DEVICES": "0,0", "CORNETTO_BG_SHARD_MIN": "1"}, {"CORNETTO_BG_PIECE": "4096"}])
def test_panel_negative_depth_values_count_as_themselves_in_the_mean(cli, tmp_path, env):
    """`%d` reads a negative depth: the reference stores its uint16 in the arrays (src/boringbits_main.c:282-283) and adds the value itself to the totals
    behind the mean (:285-286) — on one device, with the contigs dealt to several handles, with the text cut into shares and fed in pieces: stdout of the
    device path against the unmodified reference (found by tools/fuzz_cli.py, seeds 920000 and 80066 of the panel cases)"""
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "cornetto")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/cornetto not built")
    rng = np.random.default_rng(5)
    t, q = [], []
    for ci, n in enumerate((7000, 900, 4000)):
        d = rng.integers(20, 40, size=n)
        m = np.minimum(d, rng.integers(0, 40, size=n))
        for p in range(n):
            dv, mv = int(d[p]), int(m[p])
            if (ci, p) in ((0, 2767), (0, 5197), (2, 100)):
                dv = -37 if p != 5197 else -70000                    # (-70000: one more than a whole turn of the uint16 below zero)
            if (ci, p) == (1, 10):
                mv = -3
            t.append("c%d\t%d\t%d\t%d\n" % (ci, p, p + 1, dv))
            q.append("c%d\t%d\t%d\t%d\n" % (ci, p, p + 1, mv))
    a, b = tmp_path / "t.bg", tmp_path / "q.bg"
    a.write_text("".join(t))
    b.write_text("".join(q))
    for sub, opts in (("noboringbits", ["-w", "50", "-i", "1", "-m", "100", "-e", "100"]), ("boringbits", ["-w", "300", "-i", "7", "-m", "1000", "-e", "50", "-H", "1.2"]),
                      ("noboringbits", ["-w", "2500", "-i", "50", "-m", "100", "-e", "5", "-L", "0.9"])):
        args = [sub, str(a), "-q", str(b)] + opts
        pr = subprocess.run([ref] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        rc, out, err = run(cli, args, env)
        assert (rc, out) == (pr.returncode, pr.stdout), (args, env, rc, pr.returncode, len(out), len(pr.stdout), err.decode()[-300:])
        assert pr.returncode == 0 and len(pr.stdout) > 100
