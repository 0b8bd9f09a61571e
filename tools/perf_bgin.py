#!/usr/bin/env python3
"""development aid: kernel times of the device bedgraph ingest on synthetic per-base bedgraph text
   python tools/perf_bgin.py --mlines 60 --piece-mb 512"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_text(torch, dev, n, seed):
    """n lines `ptg000001l\\t%09d\\t%09d\\t%02d\\n` (34 bytes, zero padded: %d reads them the same) on the device"""
    pos = torch.arange(n, device=dev, dtype=torch.int64)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    depth = torch.randint(0, 60, (n,), device=dev, generator=g)
    out = torch.empty((n, 34), dtype=torch.uint8, device=dev)
    out[:, :10] = torch.tensor(list(b"ptg000001l"), dtype=torch.uint8, device=dev)
    out[:, 10] = 9
    out[:, 20] = 9
    out[:, 30] = 9
    out[:, 33] = 10

    def digits(v, col, nd):
        for k in range(nd):
            out[:, col + nd - 1 - k] = (v % 10 + 48).to(torch.uint8)
            v = v // 10

    digits(pos.clone(), 11, 9)
    digits(pos + 1, 21, 9)
    digits(depth.clone(), 31, 2)
    return out.reshape(-1), depth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mlines", type=float, default=60)
    ap.add_argument("--piece-mb", type=int, default=512)
    a = ap.parse_args()
    import torch
    import cornetto_amd
    dev = torch.device("cuda", 0)
    n = int(a.mlines * 1e6)
    t_dev, depth = make_text(torch, dev, n, 1)
    q_dev, depth_q = make_text(torch, dev, n, 2)
    t_host = torch.empty(t_dev.numel(), dtype=torch.uint8, pin_memory=True)
    q_host = torch.empty(q_dev.numel(), dtype=torch.uint8, pin_memory=True)
    t_host.copy_(t_dev)
    q_host.copy_(q_dev)
    torch.cuda.synchronize()
    del t_dev, q_dev
    acc = cornetto_amd.Accel(0)
    L = acc.L
    nbytes = t_host.numel()
    piece = a.piece_mb << 20
    for rep in range(2):
        bg = C.c_void_p()
        acc._chk(L.cornetto_bgin_open(acc.h, C.byref(bg)))
        ktot = {}
        t0 = time.perf_counter()
        off = 0
        while off < nbytes:
            m = min(piece, nbytes - off)
            fin = 3 if off + m >= nbytes else 0
            rc = L.cornetto_bgin_feed(acc.h, bg, C.cast(t_host.data_ptr() + off, C.c_char_p), m, C.cast(q_host.data_ptr() + off, C.c_char_p), m, fin)
            assert rc == 0, rc
            for k, ms in acc.last_timing():
                ktot[k] = ktot.get(k, 0.0) + ms
            off += m
        cov, nc, names, ncl = C.c_void_p(), C.c_int32(), C.POINTER(C.c_char_p)(), C.c_int64()
        acc._chk(L.cornetto_bgin_finish(acc.h, bg, C.byref(cov), C.byref(nc), C.byref(names), C.byref(ncl)))
        wall = time.perf_counter() - t0
        kms = sum(ktot.values())
        print("ingest: %d lines, 2 x %.2f GB text, wall %.3f s (%.2f GB/s of text incl. H2D), kernels %.1f ms (%.1f GB/s of text)" %
              (n, nbytes / 1e9, wall, 2 * nbytes / wall / 1e9, kms, 2 * nbytes / kms / 1e6), {k: round(v, 2) for k, v in ktot.items()}, flush=True)
        sums = (C.c_uint64 * 3)()
        acc._chk(L.cornetto_cov_prepare(acc.h, cov, 2500, 50, sums))
        assert int(sums[0]) == int(depth.sum().item()) and int(sums[1]) == int(depth_q.sum().item()) and int(sums[2]) == n
        L.cornetto_cov_free(acc.h, cov)
        L.cornetto_bgin_close(acc.h, bg)


if __name__ == "__main__":
    main()
