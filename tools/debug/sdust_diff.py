#!/usr/bin/env python3
"""debug aid: sdust of a FASTA through the device path against the oracle, first differences per record.
usage: python tools/debug/sdust_diff.py tests/golden/mix.fa.gz [T W]   (env: CORNETTO_SIFT_DP, CORNETTO_SDUST_CHUNK, ...)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cornetto_amd
import oracle_bind as ob
from helpers import read_fastx

path = sys.argv[1]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = int(sys.argv[3]) if len(sys.argv) > 3 else 64
recs = read_fastx(path)
acc = cornetto_amd.Accel(0)
asm = acc.asm_upload([r[2] for r in recs])
iv = acc.sdust(asm, T, W)
bad = 0
for ci, r in enumerate(recs):
    exp = [(int(x) >> 32, int(x) & 0xFFFFFFFF) for x in ob.sdust(r[2], T, W)]
    got = [(int(x["start"]), int(x["finish"])) for x in iv[iv["ctg"] == ci]]
    if exp != got:
        bad += 1
        se, sg = set(exp), set(got)
        print("record %d (%s, %d bases): %d expected, %d got; only expected %s ; only got %s" % (
            ci, r[0], len(r[2]), len(exp), len(got), sorted(se - sg)[:6], sorted(sg - se)[:6]))
print("records that differ:", bad, "of", len(recs))
