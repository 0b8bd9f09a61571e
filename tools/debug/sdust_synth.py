#!/usr/bin/env python3
"""debug aid: synthetic records through the device sdust against the oracle.  usage: sdust_synth.py [seed] [n] [T W]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cornetto_amd
import oracle_bind as ob

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 20
W = int(sys.argv[4]) if len(sys.argv) > 4 else 64
rng = np.random.default_rng(seed)
recs = []
for k in range(6):
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    for _ in range(n // 2500):
        p, L = int(rng.integers(300, n - 2000)), int(rng.integers(8, 1500))
        u = int(rng.integers(1, 8))
        unit = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=u)]
        rep = np.tile(unit, L // u + 1)[:L].copy()
        if k % 3:
            mm = rng.random(L) < (0.02 * (k % 3))
            rep[mm] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(mm.sum()))]
        s[p:p + L] = rep
    recs.append(s)
acc = cornetto_amd.Accel(0)
asm = acc.asm_upload(recs)
iv = acc.sdust(asm, T, W)
st = acc.sdust_stats(asm, T, W)
print({k: st[k] for k in ("resolve_steps", "resolve_window_reads", "dp_tiles", "after_L2")} if st else None)
bad = 0
for ci, s in enumerate(recs):
    exp = [(int(x) >> 32, int(x) & 0xFFFFFFFF) for x in ob.sdust(s, T, W)]
    got = [(int(x["start"]), int(x["finish"])) for x in iv[iv["ctg"] == ci]]
    if exp != got:
        bad += 1
        se, sg = set(exp), set(got)
        print("record %d: %d expected, %d got; only expected %s ; only got %s" % (ci, len(exp), len(got), sorted(se - sg)[:5], sorted(sg - se)[:5]))
print("records that differ:", bad, "of", len(recs))
