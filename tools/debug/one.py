import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cornetto_amd, oracle_bind as ob
seed, n = 1, 20000
rng = np.random.default_rng(seed)
recs = []
for k in range(6):
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    for _ in range(n // 2500):
        p, L = int(rng.integers(300, n - 2000)), int(rng.integers(8, 1500))
        u = int(rng.integers(1, 8))
        unit = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=u)]
        rep = np.tile(unit, L // u + 1)[:L].copy()
        if k % 3:
            mm = rng.random(L) < (0.02 * (k % 3))
            rep[mm] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(mm.sum()))]
        s[p:p + L] = rep
    recs.append(s)
s = recs[2][:14336]
acc = cornetto_amd.Accel(0)
asm = acc.asm_upload([s])
iv = acc.sdust(asm, 20, 64)
print([(int(x["start"]), int(x["finish"])) for x in iv if 12000 < x["start"] < 14000])
print([(int(x) >> 32, int(x) & 0xFFFFFFFF) for x in ob.sdust(s, 20, 64) if 12000 < (int(x) >> 32) < 14000])
