"""GPU box: build the two per-base bedgraph files of bench.py's e2e.noboringbits leg under /dev/shm and run the CLI on them with its
[INFO] timing lines shown (where the wall time of the ingest goes).  usage: python tools/ingest_probe.py [Mlines] [extra env K=V ...]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import cornetto_amd

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 100_000_000
envs = [dict(kv.split("=", 1) for kv in a.split(";") if kv) for a in sys.argv[2:]] or [{}]      # "K=V;K2=V2" per run
dev = torch.device("cuda:0")
shm = "/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp"
paths = [os.path.join(shm, "cornetto_probe_%s.%d.bg" % (t, os.getpid())) for t in ("total", "mq20")]
try:
    for path, mq in zip(paths, (False, True)):
        t = bench.make_bedgraph_text(torch, dev, n, 11, mq)
        t.cpu().numpy().tofile(path)
        del t
    torch.cuda.empty_cache()
    nbytes = sum(os.path.getsize(p) for p in paths)
    print("cores", os.cpu_count(), "text bytes", nbytes)
    for extra in envs:
        for rep in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([cornetto_amd.CLI_PATH, "noboringbits", paths[0], "-q", paths[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, CORNETTO_DEVICE="0", **extra))
            dt = time.perf_counter() - t0
            import hashlib
            print(extra, "rep", rep, "rc", p.returncode, "wall %.3f s  %.2f GB/s" % (dt, nbytes / dt / 1e9), "stdout md5", hashlib.md5(p.stdout).hexdigest())
            if rep == 1:
                print("\n".join(l for l in p.stderr.decode(errors="replace").splitlines() if "INFO" in l or "rror" in l))
finally:
    for p_ in paths:
        try:
            os.remove(p_)
        except OSError:
            pass
