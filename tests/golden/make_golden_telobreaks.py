#!/usr/bin/env python3
"""Golden vectors for `cornetto telobreaks` from the UNMODIFIED reference binary
(oracle/_ref/cornetto, built by `make -f oracle/ref.mk` from /root/reference).  Build container only; inputs and
expected outputs are committed.

    python tests/golden/make_golden_telobreaks.py
"""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "cornetto")


def run(lens, sdust, telo, out):
    with open(os.path.join(HERE, out), "wb") as fo:
        rc = subprocess.run([REF, "telobreaks", os.path.join(HERE, lens), os.path.join(HERE, sdust), os.path.join(HERE, telo)],
                            stdout=fo, stderr=subprocess.DEVNULL).returncode
    assert rc == 0, (out, rc)


def lens_from_fa2bed(src, dst):
    """scripts/telostats.sh:36: fa2bed | awk '{print $1"\\t"$3}'"""
    with open(os.path.join(HERE, src), "rb") as fi, open(os.path.join(HERE, dst), "wb") as fo:
        for line in fi:
            f = line.split()
            fo.write(f[0] + b"\t" + f[2] + b"\n")


def synthetic():
    """many contigs (hash-table growth and bucket order), every edge the reference handles without undefined
    behaviour: names missing from the lens file, a repeated lens name, short matches, uncovered flanks, contig
    edges, unsorted / overlapping / touching BED intervals"""
    rng = np.random.default_rng(20260807)
    names, lens = [], []
    for i in range(337):
        kind = i % 5
        nm = {0: "ptg%06dl" % i, 1: "h1tg%06dl" % i, 2: "chr%d_MATERNAL" % i, 3: "s%d" % i, 4: "contig_%d|arrow|pilon" % i}[kind]
        names.append(nm)
        lens.append(int(rng.integers(400, 60000)))
    lens_lines = ["%s\t%d" % (n, l) for n, l in zip(names, lens)]
    lens_lines.insert(50, "%s\t%d" % (names[7], lens[7]))          # a name twice, same length
    sd, te = [], []
    for ci, (nm, ln) in enumerate(zip(names, lens)):
        if ci % 3 == 2:
            continue                                              # contig without any record
        k = int(rng.integers(1, 6))
        for _ in range(k):
            a = int(rng.integers(0, max(1, ln - 300)))
            b = min(ln, a + int(rng.integers(150, 3000)))
            # the sdust run, cut into pieces in several ways
            style = int(rng.integers(0, 4))
            if style == 0:
                sd.append((nm, a, b))
            elif style == 1:                                      # touching halves, given in reverse order
                m = (a + b) // 2
                sd.append((nm, m, b)); sd.append((nm, a, m))
            elif style == 2:                                      # overlapping
                m = (a + b) // 2
                sd.append((nm, a, min(b, m + 40))); sd.append((nm, max(a, m - 40), b))
            else:                                                 # a one-base gap: two runs
                m = (a + b) // 2
                sd.append((nm, a, m)); sd.append((nm, m + 1, b))
            # telomere hits relative to it
            for _ in range(int(rng.integers(0, 4))):
                s = int(rng.integers(max(0, a - 150), max(1, b)))
                e = min(ln, s + 6 * int(rng.integers(1, 40)))
                if e <= s:
                    continue
                te.append((nm, ln, int(rng.integers(0, 2)), s, e, e - s))
        # hits at the very edges
        sd.append((nm, 0, min(ln, 260)))
        te.append((nm, ln, 0, 0, min(ln, 48), min(ln, 48)))
        sd.append((nm, max(0, ln - 300), ln))
        te.append((nm, ln, 1, max(0, ln - 60), ln, min(ln, 60)))
    sd.append(("not_in_lens", 5, 500)); te.append(("not_in_lens", 1000, 0, 10, 100, 90))
    te.append((names[0], lens[0], 0, 10, 33, 23))                  # matched length below MIN_TEL
    order = rng.permutation(len(sd))
    with open(os.path.join(HERE, "tb_many.lens"), "w") as f:
        f.write("\n".join(lens_lines) + "\n")
    with open(os.path.join(HERE, "tb_many.sdust"), "w") as f:
        for i in order:
            f.write("%s\t%d\t%d\n" % sd[i])
    with open(os.path.join(HERE, "tb_many.telomere"), "w") as f:
        for t in te:
            f.write("%s\t%d\t%d\t%d\t%d\t%d\n" % t)
    run("tb_many.lens", "tb_many.sdust", "tb_many.telomere", "tb_many.breaks.exp")


if __name__ == "__main__":
    lens_from_fa2bed("mix.fa2bed.exp", "mix.lens")
    run("mix.lens", "mix.sdust.exp", "mix.telofind.exp", "mix.breaks.exp")
    lens_from_fa2bed("probe.fa2bed.exp", "probe.lens")
    run("probe.lens", "probe.sdust.exp", "probe.telofind.exp", "probe.breaks.exp")
    synthetic()
    for f in ("mix.breaks.exp", "probe.breaks.exp", "tb_many.breaks.exp"):
        print(f, sum(1 for _ in open(os.path.join(HERE, f))))
