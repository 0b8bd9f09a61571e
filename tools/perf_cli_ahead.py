#!/usr/bin/env python3
"""development aid: `cornetto sdust` / `telofind` on the bench assembly as a FASTA in /dev/shm with and without the read-ahead
(CORNETTO_CLI_AHEAD), single-line and 80-column; same stdout either way"""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import cornetto_amd  # noqa: E402

EXTRA = dict(x.split('=') for x in sys.argv[1:])
dev = torch.device("cuda", 0)
lens = bench.contig_lengths(0)
bases, offs = bench.make_assembly(torch, dev, lens, 0xC0FFEE)
hb = bases.cpu().numpy()
d = "/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp"
f1, f80 = os.path.join(d, "pa1.fa"), os.path.join(d, "pa80.fa")
with open(f1, "wb") as f:
    for i, (o, L) in enumerate(zip(offs, lens)):
        f.write(b">ptg%06dl\n" % i)
        f.write(memoryview(hb[int(o):int(o) + int(L)]))
        f.write(b"\n")
with open(f80, "wb") as f:
    for i, (o, L) in enumerate(zip(offs, lens)):
        f.write(b">ptg%06dl some comment\n" % i)
        a = hb[int(o):int(o) + int(L)]
        k = len(a) // 80 * 80
        m = np.empty((k // 80, 81), dtype=np.uint8)
        m[:, :80] = a[:k].reshape(-1, 80)
        m[:, 80] = 10
        f.write(memoryview(m.reshape(-1)))
        f.write(memoryview(a[k:]))
        f.write(b"\n")
del bases, hb
torch.cuda.empty_cache()
try:
    import re
    for fa in (f1,):
        for sub in ("sdust", "telofind"):
            dig = {}
            res = {"1": [], "0": []}
            for rep in range(6):
                for ahead in ("1", "0"):
                    t0 = time.perf_counter()
                    p = subprocess.run([cornetto_amd.CLI_PATH, sub, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, CORNETTO_CLI_AHEAD=ahead, **EXTRA))
                    dt = time.perf_counter() - t0
                    dig[ahead] = hashlib.md5(p.stdout).hexdigest()
                    m = re.search(rb"Real time: ([0-9.]+) sec", p.stderr)
                    res[ahead].append((dt, float(m.group(1)) if m else -1))
            for ahead in ("1", "0"):
                w = sorted(x[0] for x in res[ahead]); r = sorted(x[1] for x in res[ahead])
                print("%-8s %-8s ahead %s: wall min %.3f median %.3f; in-process min %.3f median %.3f" % (sub, os.path.basename(fa), ahead, w[0], w[len(w) // 2], r[0], r[len(r) // 2]), flush=True)
            print("   same stdout:", dig["0"] == dig["1"])
finally:
    os.remove(f1)
    os.remove(f80)
