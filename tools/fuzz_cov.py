#!/usr/bin/env python3
"""development aid: randomised differential run of the coverage stage against the oracle with seeds the test suite does not use —
cornetto_cov_prepare (the three totals behind the mean depth), the thresholds, cornetto_cov_select / _select_packed for both selections
(print_fun_bits / print_boring_bits: src/boringbits_main.c:425-445, :463-481) over random contig sets: contigs shorter than a window, exactly a
window, one block, around the tile sizes (256 blocks, 256 windows); -w / -i with and without a remainder, -i larger than -w; depths that are
zero, flat, noisy, saturated; mq above and below the depth; thresholds around the mean; edge lengths and minimum contig lengths around the
contig lengths.  What the reference asserts on (an empty last window) is skipped the way the CLI would abort.
   python tools/fuzz_cov.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cornetto_amd  # noqa: E402
import oracle_bind as ob  # noqa: E402


def make(rng):
    w, inc = [(2500, 50), (300, 7), (1000, 1000), (2500, 49), (64, 1), (5000, 130), (777, 200), (50, 50), (50, 127), (1, 2), (2500, 4000), (97, 13),
              (256 * 50, 50), (12800, 50), (100, 100)][int(rng.integers(0, 15))]
    n_ctg = int(rng.integers(1, 9))
    lens = []
    for _ in range(n_ctg):
        kind = int(rng.integers(0, 7))
        n = int({0: rng.integers(1, max(2, w)), 1: w, 2: w + 1, 3: rng.integers(w, 4 * w + 2), 4: 256 * inc + int(rng.integers(-2, 3)),
                 5: rng.integers(20 * w, 60 * w + 2), 6: 256 * inc * int(rng.integers(1, 4)) + w + int(rng.integers(-3, 4))}[kind])
        n = max(1, min(n, 3_000_000))
        lens.append(n)
    lens = [n for n in lens if ob.regs_assert(n, w, inc) == 0] or [w]
    if ob.regs_assert(lens[0], w, inc) != 0:
        return None
    depths, mqs = [], []
    for n in lens:
        kind = int(rng.integers(0, 6))
        if kind == 0:
            d = np.zeros(n, np.uint16)
        elif kind == 1:
            d = np.full(n, int(rng.integers(1, 200)), np.uint16)
        elif kind == 2:
            d = rng.poisson(30.0, size=n).astype(np.uint16)
        elif kind == 3:
            d = (rng.poisson(30.0, size=(n + 999) // 1000).astype(np.uint16).repeat(1000)[:n] + rng.integers(0, 3, size=n).astype(np.uint16)).astype(np.uint16)
        elif kind == 4:
            d = rng.integers(0, 65536, size=n).astype(np.uint16)
        else:
            d = rng.integers(0, 4, size=n).astype(np.uint16)
        mk = int(rng.integers(0, 4))
        if mk == 0:
            q = d.copy()
        elif mk == 1:
            q = np.minimum(d, rng.integers(0, 65536, size=n)).astype(np.uint16)
        elif mk == 2:
            q = (d // 2).astype(np.uint16)
        else:
            q = rng.integers(0, 65536, size=n).astype(np.uint16)          # (an mq depth above the depth: the reference does not check)
        depths.append(d)
        mqs.append(q)
    return w, inc, lens, depths, mqs


def main():
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    acc = cornetto_amd.Accel(0)
    bad = done = 0
    n_win = n_sel = n_bases = 0
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(seed)
        m = make(rng)
        if m is None:
            continue
        w, inc, lens, depths, mqs = m
        cov = acc.cov_upload(depths, mqs)
        sums = acc.cov_prepare(cov, w, inc)
        tot = sum(int(d.astype(np.int64).sum()) for d in depths)
        totq = sum(int(q.astype(np.int64).sum()) for q in mqs)
        npos = sum(lens)
        ok = tuple(int(x) for x in sums) == (tot, totq, npos)
        mean = ob.mean_depth(tot, npos)
        lo_f, hi_f, Q = float(rng.choice([0.4, 0.5, 0.9, 0.0])), float(rng.choice([2.5, 1.1, 1.0, 4.0])), float(rng.choice([0.4, 0.0, 0.99, 1.0, 1.5]))
        lo, hi = ob.threshold(lo_f, mean), ob.threshold(hi_f, mean)
        ok = ok and (acc.cov_threshold(lo_f, mean), acc.cov_threshold(hi_f, mean)) == (lo, hi)
        edge = int(rng.choice([0, 1, w, 3 * w, 100000]))
        min_len = int(rng.choice([0, 1, lens[0], lens[0] + 1, lens[-1] - 1, 100000]))
        regs = [ob.get_regs(d, q, w, inc) for d, q in zip(depths, mqs)]
        fits16 = w <= 32768
        for boring in (False, True):
            exp = []
            for ci, (d, rs) in enumerate(zip(depths, regs)):
                nn = d.size
                if (nn > min_len) if boring else (nn >= min_len):
                    for r in rs:
                        st, end, dep, mq = int(r["st"]), int(r["end"]), int(r["depth"]), int(r["mq_depth"])
                        fun = bool(ob.is_fun(dep, mq, lo, hi, Q))
                        if (st > edge and end < nn - edge and not fun) if boring else fun:
                            exp.append((ci, st, end, dep, mq))
            got = [tuple(int(x) for x in r) for r in acc.cov_select(cov, lo, hi, Q, edge, min_len, boring)]
            ok = ok and got == exp
            n_sel += len(exp)
            if fits16:
                pk, cf = acc.cov_select_packed(cov, lo, hi, Q, edge, min_len, boring)
                un = [tuple(int(x) for x in r) for r in acc.unpack_regs(pk, cf, np.array(lens, dtype=np.int32), w)]
                ok = ok and un == exp
        # every window of every contig (cornetto_cov_regs: get_regs() itself)
        for ci in range(len(lens)):
            g = acc.cov_regs(cov, ci)
            e = regs[ci]
            ok = ok and len(g) == len(e) and all(int(a["st"]) == int(b["st"]) and int(a["end"]) == int(b["end"]) and int(a["depth"]) == int(b["depth"]) and
                                                 int(a["mq_depth"]) == int(b["mq_depth"]) for a, b in zip(g, e))
        cov.close()
        done += 1
        n_win += sum(len(r) for r in regs)
        n_bases += npos
        if not ok:
            bad += 1
            print("seed %d: MISMATCH (w %d inc %d, contigs %s, lo %d hi %d Q %g edge %d min_len %d)" % (seed, w, inc, lens, lo, hi, Q, edge, min_len), flush=True)
    acc.close()
    print("fuzz_cov: %d seeds from %d (%d run; %d positions, %d windows compared one by one, %d selected records in the two selections), %d mismatches" % (n, s0, done, n_bases, n_win, n_sel, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
