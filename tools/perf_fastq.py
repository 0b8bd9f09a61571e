#!/usr/bin/env python3
"""development aid: FASTQ text in pinned host memory -> records framed on the device -> per-read sdust
(cornetto_fastq_split + cornetto_sdust_asm), with the sequential kseq restatement of the oracle and the CLI's
host-reader path timed beside it
   python tools/perf_fastq.py --mbases 1000 --read-len 10000"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_fastq(mbases, read_len, seed):
    """ONT-like FASTQ: log-normal read lengths around read_len, uniform bases with a low-complexity stretch in every
    fourth read, qualities U[3,40]+33, header `@read%d runid=... ch=%d`"""
    rng = np.random.default_rng(seed)
    total = int(mbases * 1e6)
    lens = []
    s = 0
    while s < total:
        L = int(min(max(rng.lognormal(np.log(read_len), 0.9), 200), 200000))
        lens.append(L)
        s += L
    lens = np.array(lens, dtype=np.int64)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(lens.sum()), dtype=np.uint8)]
    quals = rng.integers(36, 74, int(lens.sum()), dtype=np.uint8)
    heads = [b"@read%d runid=5c1f3b2a9d ch=%d\n" % (i, i % 512) for i in range(len(lens))]
    size = sum(map(len, heads)) + int(2 * lens.sum()) + 4 * len(lens)
    out = np.empty(size, dtype=np.uint8)
    pos = 0
    off = 0
    for i, L in enumerate(lens.tolist()):
        h = np.frombuffer(heads[i], dtype=np.uint8)
        out[pos:pos + h.size] = h
        pos += h.size
        out[pos:pos + L] = bases[off:off + L]
        if i % 4 == 0 and L > 400:
            out[pos + 100:pos + 300] = ord("A") if i % 8 else ord("T")
        pos += L
        out[pos:pos + 3] = (10, 43, 10)
        pos += 3
        out[pos:pos + L] = quals[off:off + L]
        pos += L
        out[pos] = 10
        pos += 1
        off += L
    return out[:pos], lens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbases", type=float, default=1000)
    ap.add_argument("--read-len", type=float, default=10000)
    ap.add_argument("--min-len", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cli", action="store_true", help="also time `cornetto sdust` on the text as a file (device framing and host reader)")
    ap.add_argument("--cpu-mb", type=float, default=200, help="text given to the sequential kseq restatement")
    a = ap.parse_args()
    import cornetto_amd
    text, lens = make_fastq(a.mbases, a.read_len, 7)
    n = text.size
    L = cornetto_amd.lib()
    pin = L.cornetto_pinned_alloc(n + 64)
    C.memmove(pin, text.ctypes.data, n)
    acc = cornetto_amd.Accel(0)
    print("FASTQ text: %.1f MB, %d reads, %.1f Mbases" % (n / 1e6, len(lens), lens.sum() / 1e6), flush=True)
    for r in range(a.reps):
        t0 = time.perf_counter()
        recs, used, plain, reads = acc.fastq_split((pin, n), final=True, min_len=a.min_len, want_reads=True)
        t1 = time.perf_counter()
        k = {}
        for name, ms in acc.last_timing():
            k[name] = round(k.get(name, 0.0) + ms, 3)
        iv = acc.sdust(reads, 20, 64)
        t2 = time.perf_counter()
        ks = {}
        for name, ms in acc.last_timing():
            ks[name] = round(ks.get(name, 0.0) + ms, 3)
        kept = int(recs["keep"].sum())
        kb = int(recs["len"][recs["keep"] == 1].sum())
        assert plain and used == n and len(recs) == len(lens) and np.array_equal(recs["len"], lens.astype(np.int32))
        print("split+pack %.1f ms (%.1f GB/s of text incl. H2D; kernels %s) | sdust of %d reads / %.1f Mbases %.1f ms (%s) | "
              "text -> intervals %.1f ms = %.1f Gbases/s, %d intervals"
              % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e9, k, kept, kb / 1e6, (t2 - t1) * 1e3, {x: ks[x] for x in ks if ks[x] >= 0.05},
                 (t2 - t0) * 1e3, kb / (t2 - t0) / 1e9, len(iv)), flush=True)
        reads.close()
    import oracle_bind as ob
    m = int(min(n, a.cpu_mb * 1e6))
    t0 = time.perf_counter()
    exp, rc = ob.fastx_parse(text[:m])
    cpu = time.perf_counter() - t0
    nb = sum(len(s) for _, _, s, _ in exp)
    print("sequential kseq restatement (oracle, 1 core): %.1f MB of text in %.2f s = %.2f GB/s = %.2f Gbases/s framed"
          % (m / 1e6, cpu, m / cpu / 1e9, nb / cpu / 1e9), flush=True)
    if a.cli:
        import subprocess
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".fq", dir="/tmp", delete=True) as fh:
            fh.write(memoryview(text))
            fh.flush()
            outs = {}
            for how in ("device", "host", "device", "host"):
                e = dict(os.environ, CORNETTO_FASTQ_SPLIT=how)
                t0 = time.perf_counter()
                p = subprocess.run([cornetto_amd.CLI_PATH, "sdust", fh.name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
                dt = time.perf_counter() - t0
                outs[how] = p.stdout
                print("cornetto sdust %s (%s framing): rc %d, %.2f s wall = %.2f Gbases/s, %d bytes of output"
                      % (os.path.basename(fh.name), how, p.returncode, dt, lens.sum() / dt / 1e9, len(p.stdout)), flush=True)
            print("same output:", outs["device"] == outs["host"])
    L.cornetto_pinned_free(pin)


if __name__ == "__main__":
    main()
