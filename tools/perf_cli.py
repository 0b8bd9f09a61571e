#!/usr/bin/env python3
"""development aid: wall time of `cornetto sdust` / `cornetto telofind` on the bench's synthetic assembly written as a
single-line and as an 80-column FASTA (in /tmp, page cache), records framed on the device and by the sequential reader
   python tools/perf_cli.py"""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import cornetto_amd
    dev = torch.device("cuda", 0)
    lens = bench.contig_lengths(0)
    bases, offs = bench.make_assembly(torch, dev, lens, 0xC0FFEE)
    hb = bases.cpu().numpy()
    with open("/tmp/asm1.fa", "wb") as f:
        for i, (o, L) in enumerate(zip(offs, lens)):
            f.write(b">ptg%06dl\n" % i)
            f.write(memoryview(hb[int(o):int(o) + int(L)]))
            f.write(b"\n")
    with open("/tmp/asm80.fa", "wb") as f:
        for i, (o, L) in enumerate(zip(offs, lens)):
            f.write(b">ptg%06dl\n" % i)
            a = hb[int(o):int(o) + int(L)]
            k = len(a) // 80 * 80
            m = np.empty((k // 80, 81), dtype=np.uint8)
            m[:, :80] = a[:k].reshape(-1, 80)
            m[:, 80] = 10
            f.write(memoryview(m.reshape(-1)))
            f.write(memoryview(a[k:]))
            f.write(b"\n")
    del bases
    torch.cuda.empty_cache()
    outs = {}
    for fa in ("/tmp/asm1.fa", "/tmp/asm80.fa"):
        for sub in ("sdust", "telofind"):
            for how in ("device", "host", "device"):
                t0 = time.perf_counter()
                p = subprocess.run([cornetto_amd.CLI_PATH, sub, fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   env=dict(os.environ, CORNETTO_FASTQ_SPLIT=how))
                dt = time.perf_counter() - t0
                hsh = hashlib.md5(p.stdout).hexdigest()[:8]
                outs.setdefault(sub, set()).add(hsh)
                print(os.path.basename(fa), os.path.getsize(fa), sub, how, "framing: rc", p.returncode, "%.2f s" % dt, len(p.stdout), "bytes", hsh,
                      p.stderr[-200:] if p.returncode else "", flush=True)
    print("distinct outputs per sub-command (1 = all runs agree):", {k: len(v) for k, v in outs.items()})
    os.remove("/tmp/asm1.fa")
    os.remove("/tmp/asm80.fa")


if __name__ == "__main__":
    main()
