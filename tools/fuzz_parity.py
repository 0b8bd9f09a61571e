#!/usr/bin/env python3
"""development aid: randomised differential run of the FASTA-side scans against the oracle with seeds the test suite does not use —
telomere runs + windows (tf_scan's four-tiles-per-workgroup loop: tile counts that are and are not multiples of four, contigs shorter than a
tile, motifs of several lengths), sdust through the plain call and through cornetto_sdust_asm_begin / _end (chunk counts around the 64 counters,
other bytes, lower case, repeat arrays).
   python tools/fuzz_parity.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cornetto_amd  # noqa: E402
import oracle_bind as ob  # noqa: E402

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make(rng):
    n_ctg = int(rng.integers(1, 12))
    seqs = []
    for _ in range(n_ctg):
        kind = rng.integers(0, 6)
        ln = int({0: rng.integers(1, 80), 1: rng.integers(80, 3000), 2: rng.integers(3000, 40000), 3: rng.integers(16000, 16600),
                  4: rng.integers(60000, 70000), 5: rng.integers(100000, 400000)}[int(kind)])
        s = ACGT[rng.integers(0, 4, size=ln)].copy()
        for _ in range(int(rng.integers(0, max(2, ln // 400)))):
            p = int(rng.integers(0, ln))
            unit = [b"TTAGGG", b"CCCTAA", b"A", b"AT", b"CAG", b"N", b"acgt", b"TTAGGGTTAGGC", b"GGAAT", b"n", b"R"][int(rng.integers(0, 11))]
            rep = np.frombuffer(unit * int(rng.integers(1, 400)), dtype=np.uint8)
            seg = s[p:p + len(rep)]
            seg[:] = rep[:len(seg)]
        seqs.append(s)
    return seqs


def main():
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    acc = cornetto_amd.Accel(0)
    bad = 0
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(seed)
        seqs = make(rng)
        asm = acc.asm_upload(seqs)
        motif = [b"TTAGGG", b"CCCTAA", b"TTAGGGTTAGGG", b"GGAAT", b"AT", b"TTAGGGTTAGGGTTAGGGTT"][int(rng.integers(0, 6))]
        thr = ob.telowin_threshold(float(rng.choice([0.4, 0.1, 0.8])), float(rng.choice([99.9, 90.0])))
        hits, wins = acc.telo_scan(asm, motif, thr)
        eh, ew = [], []
        for ci, s in enumerate(seqs):
            h = ob.telofind(s.tobytes(), motif)
            eh += [(ci, int(x["strand"]), int(x["start"]), int(x["end"])) for x in h]
            w = ob.telowin(h, len(s), thr)
            ew += [(ci, int(x["start"]), int(x["end"]), int(x["car"])) for x in w]
        gh = [(int(x["ctg"]), int(x["strand"]), int(x["start"]), int(x["end"])) for x in hits]
        gw = [(int(x["ctg"]), int(x["start"]), int(x["end"]), int(x["car"])) for x in wins]
        ok_t = gh == eh and gw == ew
        T, W = [(20, 64), (25, 40), (10, 30)][int(rng.integers(0, 3))]
        es = []
        for ci, s in enumerate(seqs):
            es += [(ci, int(x) >> 32, int(x) & 0xFFFFFFFF) for x in ob.sdust(s.tobytes(), T, W)]
        ok_s = True
        for mode in ("plain", "begin/end", "begin/end", "plain"):
            if mode == "plain":
                g = acc.sdust(asm, T, W)
            else:
                acc.sdust_begin(asm, T, W)
                g = acc.sdust_end(asm, T, W)
            gs = [(int(x["ctg"]), int(x["start"]), int(x["finish"])) for x in g]
            ok_s = ok_s and gs == es
        if not (ok_t and ok_s):
            bad += 1
            print("seed %d: telo %s sdust %s (contigs %s, motif %s, T %d W %d)" % (seed, ok_t, ok_s, [len(s) for s in seqs], motif, T, W), flush=True)
        asm.close()
    acc.close()
    print("fuzz: %d seeds from %d, %d mismatches" % (n, s0, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
