#!/usr/bin/env python3
"""development aid: sdust kernel time on pure patterns (which path is slow?)"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cornetto_amd

dev = torch.device("cuda", 0)
acc = cornetto_amd.Accel(0)
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 8_000_000
rng = np.random.default_rng(5)


def run(name, arr):
    t = torch.from_numpy(arr).to(dev)
    pad = torch.zeros(256, dtype=torch.uint8, device=dev)
    t = torch.cat([t, pad])
    torch.cuda.synchronize()
    asm = acc.asm_wrap(t.data_ptr(), np.array([0], dtype=np.int64), np.array([len(arr)], dtype=np.int64))
    for _ in range(2):
        iv = acc.sdust(asm, 20, 64)
    k = dict(acc.last_timing())
    print("%-22s n=%d chunk=%s kernel %.3f ms  -> %.2f Gbases/s  ivls=%d" % (name, len(arr), os.environ.get("CORNETTO_SDUST_CHUNK", "auto"), k["sdust_kernel"], len(arr) / k["sdust_kernel"] / 1e6, len(iv)), flush=True)
    asm.close()


acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
rand = acgt[rng.integers(0, 4, size=n)]
run("random", rand)
run("telomere CCCTAA", np.frombuffer(b"CCCTAA" * (n // 6), dtype=np.uint8).copy())
run("homopolymer A", np.full(n, ord("A"), dtype=np.uint8))
run("dinuc AC", np.frombuffer(b"AC" * (n // 2), dtype=np.uint8).copy())
x = rand.copy()
x[::1000] = ord("N")
run("random + N every 1k", x)
x = rand.copy()
for p in range(0, n - 20000, 500000):
    x[p:p + 12000] = np.frombuffer(b"CCCTAA" * 2000, dtype=np.uint8)
run("random + 12k telo/500k", x)
