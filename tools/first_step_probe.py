import sys, os, time, argparse
sys.path.insert(0, os.getcwd())
import bench, torch, torch.distributed as dist, cornetto_amd, numpy as np
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1"]
ap_args = None
# reuse bench's parser by calling main's pieces
import types
args = types.SimpleNamespace(gpus=1, steps=3, warmup=1, gbases=0.0, scaling="weak", profile="uniform", assembly_index=0, timing=1, serial=False, sdust_share=-1, gather=False,
                             allreduce_always=False, split_tol=0.05, allow_shared_device=False)
R = bench.Rank(args, torch, dist, cornetto_amd)
R.load("uniform")
for k in range(3):
    R.wall = {}
    t0 = time.perf_counter()
    R.step(True)
    print("step %d: %.2f ms" % (k, (time.perf_counter() - t0) * 1e3), {a: round(b[0], 2) for a, b in R.wall.items()}, flush=True)
R.wrap(R.own)
for k in range(2):
    R.wall = {}
    t0 = time.perf_counter()
    R.step(True)
    print("re-wrapped step %d: %.2f ms" % (k, (time.perf_counter() - t0) * 1e3), {a: round(b[0], 2) for a, b in R.wall.items()}, flush=True)
R.unload(); R.close()
