#!/usr/bin/env python3
"""development aid: where a step of a 1/N share of the bench assembly spends its time — every API call alone (wall) with the
per-launch HIP events (--timing 2), then the two-thread step.   python tools/perf_share.py [N=8] [reps=20]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import cornetto_amd  # noqa: E402
from cornetto_amd.dist import lpt_partition  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
lens = bench.contig_lengths(0)
bases, offs = bench.make_assembly(torch, dev, lens, 0xC0FFEE)
depth, mq = bench.make_coverage(torch, dev, lens, offs, 0xC0FFEE)
torch.cuda.synchronize()
parts = lpt_partition(lens, n)
own = max(parts, key=lambda p: sum(lens[i] for i in p))
o = offs[own]
ln = np.array([lens[i] for i in own], dtype=np.int64)
acc = cornetto_amd.Accel(0)
asm = acc.asm_wrap(bases.data_ptr(), o, ln)
cov = acc.cov_wrap(depth.data_ptr(), mq.data_ptr(), o, ln.astype(np.int32))
thr = acc.telowin_threshold(0.4, 99.9)
print("share: %d contigs, %.1f Mbases" % (len(own), ln.sum() / 1e6))
if os.environ.get("PERF_LAZY") == "1":        # result copies on the handle's copy stream, met at wait() (what the bench's second thread does)
    acc.set_lazy(True)


def timed(name, fn):
    fn()
    fn()
    acc.set_timing(0)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        acc.wait()
    wall = (time.perf_counter() - t0) / reps * 1e3
    acc.set_timing(2)
    fn()
    k = acc.last_timing()
    acc.set_timing(0)
    print("%-12s wall %.3f ms; kernels %.3f ms in %d launches: %s" % (name, wall, sum(ms for _, ms in k), len(k), " ".join("%s %.3f" % (a, b) for a, b in k)))


timed("sdust", lambda: acc.sdust(asm, 20, 64))
timed("telo_scan", lambda: acc.telo_scan(asm, b"TTAGGG", thr))
timed("cov_prepare", lambda: acc.cov_prepare(cov, 2500, 50))
sd, sq, nn = acc.cov_prepare(cov, 2500, 50)
mean = int(np.floor(sd / nn + 0.5))
lo, hi = acc.cov_threshold(0.4, mean), acc.cov_threshold(2.5, mean)
timed("cov_select", lambda: acc.cov_select_packed(cov, lo, hi, 0.4, 100000, 1000000, False))
