"""GPU box: the synthetic 3.16 Gbp assembly as a FASTA file under /dev/shm, `cornetto sdust` on it with the phase trace, for a few
environments.  usage: python tools/fasta_probe.py [gbases] ["K=V;K2=V2" ...]"""
import hashlib
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import cornetto_amd

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 3.16
envs = [dict(kv.split("=", 1) for kv in a.split(";") if kv) for a in sys.argv[2:]] or [{}]
dev = torch.device("cuda:0")
lens = bench.contig_lengths(int(gb * 1e9))
bases, offs = bench.make_assembly(torch, dev, lens, 0xC0FFEE, "uniform")
hb = bases.cpu().numpy()
path = os.path.join(bench._shm_dir(), "cornetto_probe.%d.fa" % os.getpid())
try:
    with open(path, "wb") as f:
        for i, n in enumerate(lens):
            f.write(b">ptg%06dl\n" % i)
            f.write(memoryview(hb[int(offs[i]):int(offs[i]) + int(n)]))
            f.write(b"\n")
    del hb, bases
    torch.cuda.empty_cache()
    print("fasta bytes", os.path.getsize(path), "cores", os.cpu_count())
    for extra in envs:
        for rep in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([cornetto_amd.CLI_PATH, "sdust", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, CORNETTO_DEVICE="0", CORNETTO_CLI_TRACE="1", **extra))
            dt = time.perf_counter() - t0
            tr = [l[len("[cli trace]"):].strip() for l in p.stderr.decode(errors="replace").splitlines() if l.startswith("[cli trace]")]
            print(extra, "rep", rep, "rc", p.returncode, "wall %.3f s" % dt, "md5", hashlib.md5(p.stdout).hexdigest()[:12], "|", "; ".join(tr[:3] + tr[-2:]))
finally:
    try:
        os.remove(path)
    except OSError:
        pass
