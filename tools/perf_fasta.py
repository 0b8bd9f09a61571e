#!/usr/bin/env python3
"""development aid: FASTA text (the bench's synthetic assembly, 80-column or single-line) in pinned host memory ->
records framed and sequences laid out on the device (cornetto_fasta_split), per-kernel times
   python tools/perf_fasta.py --width 80 --mbases 3160"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbases", type=float, default=3160)
    ap.add_argument("--width", type=int, default=80, help="0 = one line per record")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import bench
    import cornetto_amd
    dev = torch.device("cuda", 0)
    lens = bench.contig_lengths(int(a.mbases * 1e6))
    bases, offs = bench.make_assembly(torch, dev, lens, 0xC0FFEE)
    hb = bases.cpu().numpy()
    parts = []
    for i, (o, L) in enumerate(zip(offs, lens)):
        s = hb[int(o):int(o) + int(L)]
        parts.append(np.frombuffer(b">ptg%06dl\n" % i, dtype=np.uint8))
        if a.width:
            k = len(s) // a.width * a.width
            m = np.empty((k // a.width, a.width + 1), dtype=np.uint8)
            m[:, :a.width] = s[:k].reshape(-1, a.width)
            m[:, a.width] = 10
            parts += [m.reshape(-1), s[k:], np.frombuffer(b"\n", dtype=np.uint8)]
        else:
            parts += [s, np.frombuffer(b"\n", dtype=np.uint8)]
    text = np.concatenate(parts)
    n = text.size
    L = cornetto_amd.lib()
    pin = L.cornetto_pinned_alloc(n + 64)
    C.memmove(pin, text.ctypes.data, n)
    acc = cornetto_amd.Accel(0)
    print("FASTA text: %.1f MB, %d records, %.1f Mbases, %s" % (n / 1e6, len(lens), sum(lens) / 1e6, "%d-column" % a.width if a.width else "single-line"), flush=True)
    for r in range(a.reps):
        t0 = time.perf_counter()
        recs, used, plain, seqs = acc.fasta_split((pin, n), final=True, want_seqs=True)
        dt = time.perf_counter() - t0
        k = {}
        for name, ms in acc.last_timing():
            k[name] = round(k.get(name, 0.0) + ms, 3)
        assert plain and used == n and [int(x) for x in recs["len"]] == [int(x) for x in lens]
        print("split %.1f ms = %.1f GB/s of text incl. H2D; kernels %.2f ms %s" % (dt * 1e3, n / dt / 1e9, sum(k.values()), k), flush=True)
        if r == a.reps - 1:
            # the laid-out sequences are the assembly: same sdust intervals as on the original device buffer
            asm0 = acc.asm_wrap(bases.data_ptr(), offs, np.array(lens, dtype=np.int64))
            i0, i1 = acc.sdust(asm0, 20, 64), acc.sdust(seqs, 20, 64)
            print("sdust over the framed sequences == over the source assembly:", bool(np.array_equal(i0, i1)), len(i1), flush=True)
            asm0.close()
        seqs.close()
    L.cornetto_pinned_free(pin)


if __name__ == "__main__":
    main()
