#!/usr/bin/env python3
"""perf probe (development aid, not a test): kernel times of one stage for a synthetic assembly.
   python tools/perf_probe.py sdust --mbases 500 --features 0 --chunk 4096"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stage", choices=["sdust", "telo", "cov", "all"])
    ap.add_argument("--mbases", type=float, default=500)
    ap.add_argument("--features", type=int, default=1)
    ap.add_argument("--chunk", type=str, default="")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--profile", default="uniform", help="bench.make_assembly profile: uniform | satellite")
    ap.add_argument("--read-len", type=int, default=0, help="instead of the assembly: reads of about this length (+-30 %%), --mbases in total")
    ap.add_argument("--simple-cov", type=int, default=0, help="uniform random depth (rocprofv3 --pmc crashes inside torch.poisson)")
    a = ap.parse_args()
    if a.chunk:                     # (a development switch: the development build of the library reads it)
        os.environ["CORNETTO_SDUST_CHUNK"] = a.chunk
        os.environ.setdefault("CORNETTO_LIB", os.path.join(ROOT, "cornetto_amd", "libcornetto_hip_dev.so"))
    import torch
    import bench
    import cornetto_amd
    dev = torch.device("cuda", 0)
    lens = bench.contig_lengths(int(a.mbases * 1e6))
    if a.read_len > 0:
        rng = np.random.default_rng(7)
        lens = [int(x) for x in rng.integers(int(a.read_len * 0.7), int(a.read_len * 1.3), size=int(a.mbases * 1e6 / a.read_len))]
        a.features = 0
    if a.features:
        bases, offs = bench.make_assembly(torch, dev, lens, 0xC0FFEE if a.profile != "uniform" else 1, a.profile)
    else:
        offs, pos = [], 0
        for n in lens:
            offs.append(pos)
            pos = (pos + n + 63) // 64 * 64
        offs = np.array(offs, dtype=np.int64)
        lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        bases = lut[torch.randint(0, 4, (pos + 256,), device=dev)]
    torch.cuda.synchronize()
    acc = cornetto_amd.Accel(0)
    asm = acc.asm_wrap(bases.data_ptr(), offs, np.array(lens, dtype=np.int64))
    n = sum(lens)
    for rep in range(a.reps):
        if a.stage in ("sdust", "all"):
            import time
            t0 = time.perf_counter()
            iv = acc.sdust(asm, 20, 64)
            print("sdust call %.1f ms;" % ((time.perf_counter() - t0) * 1e3), len(lens), "records;", end=" ")
            print("sdust", a.mbases, "features", a.features, "chunk", a.chunk, [(k, round(v, 3)) for k, v in acc.last_timing()], "ivls", len(iv), "digest", bench.digest([iv]), flush=True)
        if a.stage in ("telo", "all"):
            h, w = acc.telo_scan(asm, b"TTAGGG", 0.3976)
            print("telo", [(k, round(v, 3)) for k, v in acc.last_timing()], len(h), len(w), flush=True)
        if a.stage in ("cov", "all"):
            if a.simple_cov:
                tot = int(offs[-1] + (lens[-1] + 63) // 64 * 64 + 256)
                depth = torch.randint(0, 60, (tot,), dtype=torch.int16, device=dev)
                mq = depth.clone()
            else:
                depth, mq = bench.make_coverage(torch, dev, lens, offs, 1)
            torch.cuda.synchronize()
            cov = acc.cov_wrap(depth.data_ptr(), mq.data_ptr(), offs, np.array(lens, dtype=np.int32))
            s = acc.cov_prepare(cov, 2500, 50)
            print("cov_prepare", [(k, round(v, 3)) for k, v in acc.last_timing()], flush=True)
            r = acc.cov_select(cov, 12, 75, 0.4, 100000, 1000000, False)
            print("cov_select", [(k, round(v, 3)) for k, v in acc.last_timing()], len(r), flush=True)
            import time
            t0 = time.perf_counter()
            r = acc.cov_select(cov, 12, 75, 0.4, 100000, 1000000, False)
            t1 = time.perf_counter()
            m = acc.cov_select_merged(cov, 12, 75, 0.4, 100000, 1000000, False, 1000, 30000)
            t2 = time.perf_counter()
            print("cov_select call %.2f ms (%d windows to the host); cov_select_merged call %.2f ms (%d intervals)" % ((t1 - t0) * 1e3, len(r), (t2 - t1) * 1e3, len(m)),
                  [(k, round(v, 3)) for k, v in acc.last_timing()], flush=True)


if __name__ == "__main__":
    main()
