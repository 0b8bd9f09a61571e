#!/usr/bin/env python3
"""development aid (build container, no GPU): the oracle (oracle/oracle.c through tests/oracle_bind.py) against the unmodified reference binary on random
FASTA text — the third side of the triangle the GPU tests stand on (device == oracle: tests -m gpu, tools/fuzz_parity.py, fuzz_cov.py; CLI == reference:
tools/fuzz_cli.py): telofind with several motifs, sdust with several (T, W), telowin on the reference's telofind rows; stdout formatted with the
reference's printf formats (tests/helpers.py).
   python tools/fuzz_oracle.py [first_seed] [n_seeds]"""
import os
import random
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_cli  # noqa: E402
import helpers  # noqa: E402
import oracle_bind as ob  # noqa: E402


def main():
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "o.fa")
        for seed in range(s0, s0 + n):
            rnd = random.Random(seed)
            text = fuzz_cli.fasta_text(rnd)
            open(path, "wb").write(text)
            recs = helpers.read_fastx(path)                         # (the oracle's own framing restatement: orc_fastx_parse)
            motif = rnd.choice([b"TTAGGG", b"CCCTAA", b"TTTAGGG", b"AAAA", b"ACACA", b"AC", b"TTAGGGTTAGGG", b"ACGT" * 9, b"GGAAT"])
            exp = fuzz_cli.run(fuzz_cli.REF, ["telofind", path, motif.decode()])[1]
            got = b"".join(helpers.fmt_telofind(nm, len(sq), ob.telofind(np.frombuffer(sq, dtype=np.uint8), motif)) for nm, _c, sq, _q in recs)
            ok_t = got == exp
            T, W = rnd.choice([(20, 64), (25, 40), (10, 30), (5, 16), (30, 100), (2, 8)])
            exp = fuzz_cli.run(fuzz_cli.REF, ["sdust", "-w", str(W), "-t", str(T), path])[1]
            got = b"".join(helpers.fmt_sdust(nm, ob.sdust(np.frombuffer(sq, dtype=np.uint8), T, W)) for nm, _c, sq, _q in recs)
            ok_s = got == exp
            if not (ok_t and ok_s):
                bad += 1
                print("seed %d: telofind %s (motif %s), sdust %s (T %d, W %d)" % (seed, ok_t, motif, ok_s, T, W), flush=True)
    print("fuzz_oracle: %d seeds from %d, %d mismatches" % (n, s0, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
