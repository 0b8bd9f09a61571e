"""telobreaks: the khash bucket order on the host (no GPU), the device bitset stage
against the oracle and the reference's golden stdout, and the CLI end to end."""
import os
import subprocess

import numpy as np
import pytest

import cornetto_amd
import oracle_bind as ob
from helpers import golden
from test_oracle_golden import TELOBREAKS_CASES, telobreaks_text


@pytest.fixture(scope="module")
def acc():
    a = cornetto_amd.Accel(0)
    yield a
    a.close()


def _names(rng, n):
    shapes = [lambda i: b"ptg%06dl" % i, lambda i: b"chr%d" % i, lambda i: b"h2tg%06dc" % (i * 31), lambda i: b"x" * (1 + i % 40) + b"%d" % i,
              lambda i: bytes([0xC3, 0xA9]) + b"_%d" % i]           # a byte >= 0x80: the X31 hash works on signed char
    return [shapes[int(rng.integers(0, len(shapes)))](int(rng.integers(0, max(2, n // 2)))) for _ in range(n)]


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 12, 13, 100, 789, 5000])
def test_khash_order_matches_oracle(n):
    """cornetto_khash_str_order (product, host only) vs the oracle's restatement, duplicates included"""
    rng = np.random.default_rng(n)
    names = _names(rng, n)
    s1, o1 = cornetto_amd.khash_str_order(names)
    s2, o2 = ob.khash_order(names)
    assert np.array_equal(s1, s2) and np.array_equal(o1, o2)
    assert len(o1) == len(set(names))


def _to_product(res):
    out = np.zeros(len(res), cornetto_amd.IVL_DT)
    out["ctg"], out["start"], out["finish"] = res["ctg"], res["start"], res["end"]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("lens_f,sd_f,tel_f,exp", TELOBREAKS_CASES)
def test_telobreaks_golden(acc, golden_dir, lens_f, sd_f, tel_f, exp):
    def breaks(ctg_len, sd, tel):
        sd_p = np.zeros(len(sd), cornetto_amd.IVL_DT)
        sd_p["ctg"], sd_p["start"], sd_p["finish"] = sd["ctg"], sd["start"], sd["end"]
        r = acc.telobreaks(ctg_len, sd_p, tel.astype(cornetto_amd.TELROW_DT))
        out = np.zeros(len(r), ob.SPAN_DT)
        out["ctg"], out["start"], out["end"] = r["ctg"], r["start"], r["finish"]
        return out
    assert telobreaks_text(golden_dir, lens_f, sd_f, tel_f, breaks, cornetto_amd.khash_str_order) == golden(golden_dir, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_ctg,maxlen", [(1, 1, 70), (2, 7, 200), (3, 40, 5000), (4, 300, 70000), (5, 3, 2_000_000)])
def test_telobreaks_random_vs_oracle(acc, seed, n_ctg, maxlen):
    """random contigs, interval soups and telomere rows (word edges, contig edges, one-bit gaps, long runs)"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, maxlen + 1, size=n_ctg).astype(np.int32)
    lens[0] = maxlen
    sd, tel = [], []
    for c, ln in enumerate(lens):
        for _ in range(int(rng.integers(0, 12))):
            a = int(rng.integers(0, ln))
            b = min(int(ln), a + int(rng.integers(1, max(2, ln // 2))))
            cuts = sorted(set([a, b] + [int(x) for x in rng.integers(a, b + 1, size=int(rng.integers(0, 4)))]))
            for x, y in zip(cuts[:-1], cuts[1:]):
                if rng.random() < 0.85:
                    sd.append((c, x, y))
                else:
                    sd.append((c, x, max(x, y - 1)))              # a one-base hole (or an empty interval)
            for _ in range(int(rng.integers(0, 6))):
                s = int(rng.integers(max(0, a - 130), b))
                e = min(int(ln), s + int(rng.integers(1, 300)))
                if e > s:
                    tel.append((c, s, e, int(rng.choice([e - s, 23, 24, 5]))))
        if ln >= 64:
            sd.append((c, 0, 64)); tel.append((c, 0, 30, 30))      # exactly one word
            sd.append((c, int(ln) - 64, int(ln))); tel.append((c, int(ln) - 40, int(ln), 40))
    sd.append((-1, 0, 10)); tel.append((-1, 0, 30, 30))            # rows of names that are not in the lens file
    rng.shuffle(sd)
    sd_o = np.array(sd, dtype=ob.SPAN_DT).reshape(-1)
    tel_o = np.array(tel, dtype=ob.TELROW_DT).reshape(-1)
    exp = ob.telobreaks(lens, sd_o, tel_o)
    sd_p = np.zeros(len(sd_o), cornetto_amd.IVL_DT)
    sd_p["ctg"], sd_p["start"], sd_p["finish"] = sd_o["ctg"], sd_o["start"], sd_o["end"]
    got = acc.telobreaks(lens, sd_p, tel_o.astype(cornetto_amd.TELROW_DT))
    assert np.array_equal(got, _to_product(exp)), (len(got), len(exp))


@pytest.mark.gpu
def test_telobreaks_rejects_coordinates_outside_the_contig(acc):
    """unchecked heap indices in the reference (src/telomere_breaks.c:86,:106): a clean error here, as in the oracle"""
    lens = np.array([1000], np.int32)
    ok_sd = np.array([(0, 0, 500)], cornetto_amd.IVL_DT)
    ok_tel = np.array([(0, 100, 200, 100)], cornetto_amd.TELROW_DT)
    assert len(acc.telobreaks(lens, ok_sd, ok_tel)) == 1
    # an sdust interval that ends beyond the contig (sdust prints them at a contig's end: up to W beyond the last base) is cut at the end, which is
    # what the reference's unchecked writes amount to (:85 sets bits that :103,:118,:136 never read)
    for fin in (1001, 1063, 5000):
        sd_over = np.array([(0, 700, fin)], cornetto_amd.IVL_DT)
        tel_end = np.array([(0, 900, 1000, 100)], cornetto_amd.TELROW_DT)
        got = acc.telobreaks(lens, sd_over, tel_end)
        assert [tuple(int(x) for x in r) for r in got] == [(0, 699, 999)]
        exp = ob.telobreaks(lens, np.array([(0, 700, fin)], ob.SPAN_DT), tel_end.astype(ob.TELROW_DT))
        assert [tuple(int(x) for x in r) for r in exp] == [(0, 699, 999)]
    for sd, tel in ((ok_sd, np.array([(0, 900, 1001, 101)], cornetto_amd.TELROW_DT)),
                    (np.array([(0, -1, 5)], cornetto_amd.IVL_DT), ok_tel)):
        with pytest.raises(cornetto_amd.AccelError):
            acc.telobreaks(lens, sd, tel)
        assert ob.telobreaks(lens, np.array([(r["ctg"], r["start"], r["finish"]) for r in sd], ob.SPAN_DT), tel.astype(ob.TELROW_DT)) is None


@pytest.mark.gpu
@pytest.mark.parametrize("lens_f,sd_f,tel_f,exp", TELOBREAKS_CASES)
def test_telobreaks_cli_golden(golden_dir, lens_f, sd_f, tel_f, exp):
    p = subprocess.run([cornetto_amd.CLI_PATH, "telobreaks"] + [os.path.join(golden_dir, f) for f in (lens_f, sd_f, tel_f)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr[-400:]
    assert p.stdout == golden(golden_dir, exp)


def test_telobreaks_cli_usage_and_missing_file(tmp_path):
    """exit codes of src/telomere_breaks.c:48-51 and F_CHK (:60) — decided before any device is touched"""
    if not os.path.exists(cornetto_amd.CLI_PATH):
        pytest.skip("CLI not built")
    p = subprocess.run([cornetto_amd.CLI_PATH, "telobreaks", "a", "b"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"Usage: telobreaks" in p.stderr and p.stdout == b""
    p = subprocess.run([cornetto_amd.CLI_PATH, "telobreaks", str(tmp_path / "none.lens"), "b", "c"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and p.stdout == b""
