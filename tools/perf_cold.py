#!/usr/bin/env python3
"""the COLD pass (development aid): every stage of a step called once on a fresh handle over a freshly wrapped resident assembly — wall time of the
first, second and third call of each entry point beside the sum of its kernels' times: what the first (and for a panel run: only) pass over an
assembly pays for tables, workspaces and pinned buffers.   python tools/perf_cold.py [--gbases 0] [--rounds 3]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gbases", type=float, default=0.0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--fresh", type=int, default=2, help="how many fresh handle + wrap cycles")
    ap.add_argument("--warm", type=int, default=0, help="1: every entry point once on a 4 kb assembly before anything is timed (what a warm-up inside cornetto_accel_open would buy)")
    ap.add_argument("--dev", type=int, default=0, help="1: the development build of the library (CORNETTO_SDUST_TRACE=1 prints the phases of a call)")
    a = ap.parse_args()
    import torch
    import cornetto_amd
    from cornetto_amd import synth
    dev = torch.device("cuda", 0)
    lens = synth.contig_lengths(int(a.gbases * 1e9) if a.gbases > 0 else 0)
    bases, offs = synth.make_assembly(torch, dev, lens, 0xC0FFEE)
    depth, mq = synth.make_coverage(torch, dev, lens, offs, 0xC0FFEE)
    torch.cuda.synchronize()
    ln64, ln32 = np.array(lens, dtype=np.int64), np.array(lens, dtype=np.int32)
    if a.warm:
        t0 = time.perf_counter()
        wa = cornetto_amd.Accel(0, dev=bool(a.dev))
        rng = np.random.default_rng(1)
        sq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 4096)].copy()
        sq[100:400] = np.frombuffer(b"TTAGGG" * 50, dtype=np.uint8)
        wasm = wa.asm_upload([sq])
        wcov = wa.cov_upload([rng.integers(10, 50, 4096).astype(np.uint16)], [rng.integers(10, 50, 4096).astype(np.uint16)])
        wa.sdust(wasm, 20, 64); wa.telo_scan(wasm, b"TTAGGG", 0.39); wa.cov_prepare(wcov, 2500, 50); wa.cov_select_packed(wcov, 10, 60, 0.4, 100, 1000, False)
        wasm.close(); wcov.close(); wa.close()
        print("warm-up on 4 kb: %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    for f in range(a.fresh):
        t0 = time.perf_counter()
        acc = cornetto_amd.Accel(0, dev=bool(a.dev))
        acc.set_timing(2)
        t_open = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        asm = acc.asm_wrap(bases.data_ptr(), offs, ln64)
        cov = acc.cov_wrap(depth.data_ptr(), mq.data_ptr(), offs, ln32)
        t_wrap = (time.perf_counter() - t0) * 1e3
        print("fresh %d: open %.2f ms, wrap %.2f ms" % (f, t_open, t_wrap), flush=True)
        thr = acc.telowin_threshold(0.4, 99.9)
        for r in range(a.rounds):
            row = []
            t0 = time.perf_counter()
            iv = acc.sdust(asm, 20, 64)
            row.append(("sdust", (time.perf_counter() - t0) * 1e3, sum(v for _, v in acc.last_timing())))
            t0 = time.perf_counter()
            sums = acc.cov_prepare(cov, 2500, 50)
            row.append(("cov_prepare", (time.perf_counter() - t0) * 1e3, sum(v for _, v in acc.last_timing())))
            mean = int(np.floor(sums[0] / sums[2] + 0.5))
            lo, hi = acc.cov_threshold(0.4, mean), acc.cov_threshold(2.5, mean)
            t0 = time.perf_counter()
            pk, cf = acc.cov_select_packed(cov, lo, hi, 0.4, 100000, 1000000, False)
            row.append(("cov_select", (time.perf_counter() - t0) * 1e3, sum(v for _, v in acc.last_timing())))
            t0 = time.perf_counter()
            h, w = acc.telo_scan(asm, b"TTAGGG", thr)
            row.append(("telo_scan", (time.perf_counter() - t0) * 1e3, sum(v for _, v in acc.last_timing())))
            print("  round %d: " % r + "; ".join("%s %.2f ms (kernels %.2f)" % x for x in row) + "; total %.2f ms" % sum(x[1] for x in row), flush=True)
            del iv, pk, cf, h, w
        asm.close()
        cov.close()
        acc.close()


if __name__ == "__main__":
    main()
