#!/usr/bin/env python3
"""development aid: device time of the telobreaks bitset stage on the bench's synthetic assembly (its own sdust
intervals and telofind rows as input), with the CPU oracle timed on the first contigs for comparison
   python tools/perf_telobreaks.py --mbases 3160"""
import argparse
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
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cpu-contigs", type=int, default=3)
    a = ap.parse_args()
    import torch
    import bench
    import cornetto_amd
    dev = torch.device("cuda", 0)
    lens = bench.contig_lengths(int(a.mbases * 1e6))
    bases, offs = bench.make_assembly(torch, dev, lens, 1234)
    torch.cuda.synchronize()
    acc = cornetto_amd.Accel(0)
    asm = acc.asm_wrap(bases.data_ptr(), offs, np.array(lens, dtype=np.int64))
    iv = acc.sdust(asm, 20, 64)
    hits, _wins = acc.telo_scan(asm, b"TTAGGG", acc.telowin_threshold(0.4, 99.9))
    tel = np.zeros(len(hits), cornetto_amd.TELROW_DT)
    tel["ctg"], tel["start"], tel["end"] = hits["ctg"], hits["start"], hits["end"]
    tel["matched"] = hits["end"] - hits["start"]
    ctg_len = np.array(lens, dtype=np.int32)
    n_bases = int(ctg_len.sum())
    for r in range(a.reps):
        t0 = time.perf_counter()
        res = acc.telobreaks(ctg_len, iv, tel)
        wall = time.perf_counter() - t0
        k = {}
        for name, ms in acc.last_timing():
            k[name] = round(k.get(name, 0.0) + ms, 4)
        dev_ms = sum(k.values())
        print("telobreaks: %d intervals, %d telomere rows (%d of >= 24) -> %d runs; kernels %.3f ms (%.0f Gbases/s) %s; call %.1f ms incl. H2D of the rows"
              % (len(iv), len(tel), int((tel["matched"] >= 24).sum()), len(res), dev_ms, n_bases / dev_ms / 1e6, k, wall * 1e3), flush=True)
    # CPU oracle on the leading contigs
    import oracle_bind as ob
    nc = min(a.cpu_contigs, len(lens))
    sd_o = np.zeros(int((iv["ctg"] < nc).sum()), ob.SPAN_DT)
    m = iv["ctg"] < nc
    sd_o["ctg"], sd_o["start"], sd_o["end"] = iv["ctg"][m], iv["start"][m], iv["finish"][m]
    tel_o = tel[tel["ctg"] < nc].astype(ob.TELROW_DT)
    t0 = time.perf_counter()
    exp = ob.telobreaks(ctg_len[:nc], sd_o, tel_o)
    cpu = time.perf_counter() - t0
    got = res[res["ctg"] < nc]
    same = len(got) == len(exp) and np.array_equal(got["start"], exp["start"]) and np.array_equal(got["finish"], exp["end"])
    print("oracle (1 core) on the first %d contigs, %d bases: %.2f s = %.3f Gbases/s; same runs as the device: %s"
          % (nc, int(ctg_len[:nc].sum()), cpu, ctg_len[:nc].sum() / cpu / 1e9, same), flush=True)


if __name__ == "__main__":
    main()
