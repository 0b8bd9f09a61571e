#!/usr/bin/env python3
"""development aid: the drop-in sdust entry points of the C ABI — cornetto_sdust() and cornetto_sdust_buf_init() / cornetto_sdust_core() /
cornetto_sdust_buf_destroy(), include/cornetto_accel.h — against the UNMODIFIED reference's sdust() (src/sdust/sdust.c:162-171, from
oracle/_ref/libcornetto_ref.so, which travels with the snapshot) in one process: the same bytes, l_seq given and -1 (strlen), T and W varied
over what the reference accepts, sequences of every byte value (no NUL when the length is -1), lengths 0 .. 200 kb, low-complexity runs at
both ends (intervals that reach beyond the sequence), lower case, other letters.  The result arrays must be equal word for word.
   python tools/fuzz_abi_sdust.py [first_seed] [n_seeds]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cornetto_amd  # noqa: E402


def main():
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    refp = os.path.join(ROOT, "oracle", "_ref", "libcornetto_ref.so")
    if not os.path.exists(refp):
        print("oracle/_ref/libcornetto_ref.so is not built")
        return 2
    R = C.CDLL(refp)
    R.sdust.restype = C.POINTER(C.c_uint64)
    R.sdust.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L = cornetto_amd.lib()
    L.cornetto_sdust.restype = C.POINTER(C.c_uint64)
    L.cornetto_sdust.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.cornetto_sdust_buf_init.restype = C.c_void_p
    L.cornetto_sdust_buf_init.argtypes = [C.c_void_p]
    L.cornetto_sdust_buf_destroy.argtypes = [C.c_void_p]
    L.cornetto_sdust_core.restype = C.POINTER(C.c_uint64)
    L.cornetto_sdust_core.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    buf = L.cornetto_sdust_buf_init(None)
    bad = words = bases = 0
    units = [b"A", b"AT", b"CATTC", b"AAAG", b"TTAGGG", b"GGAAT", b"acgt", b"N", b"CAG", b"ttaggg"]
    for seed in range(s0, s0 + n):
        rng = np.random.default_rng(seed)
        ln = int(rng.choice([0, 1, 2, 3, 7, 63, 64, 65, 127, 200, 1000, 5000, 20000])) if rng.random() < 0.5 else int(rng.integers(0, 200001 if rng.random() < 0.1 else 30001))
        kind = rng.random()
        if kind < 0.5:
            s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=ln)].copy()
        elif kind < 0.75:
            s = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)[rng.integers(0, 10, size=ln)].copy()
        else:
            s = rng.integers(1, 256, size=ln, dtype=np.uint8)
        for _ in range(int(rng.integers(0, max(2, ln // 250)))):
            if ln < 8:
                break
            p = int(rng.integers(0, ln))
            rep = np.frombuffer(units[int(rng.integers(0, len(units)))] * int(rng.integers(1, 300)), dtype=np.uint8)
            seg = s[p:p + len(rep)]
            seg[:] = rep[:len(seg)]
        if ln >= 40 and rng.random() < 0.4:                 # low-complexity runs flush with both ends: intervals beyond the sequence
            s[:30] = ord("A")
            s[ln - 35:] = np.frombuffer(b"TTAGGG" * 6, dtype=np.uint8)[:35]
        T = int(rng.choice([20, 20, 25, 10, 5, 30, 2, 64, 100]))
        W = int(rng.choice([64, 64, 40, 30, 16, 8, 4, 66, 100, 200, 500, 1026]))
        if W > 100 and ln > (400 if W > 200 else 3000):      # (the reference's find_perfect is quadratic in W inside a repeat: seconds per kilobase at W = 1026)
            s = s[:400 if W > 200 else 3000].copy()
            ln = len(s)
        by_strlen = ln > 0 and rng.random() < 0.3 and not np.any(s == 0)
        cbuf = C.create_string_buffer(s.tobytes(), ln + 1)
        l_arg = -1 if by_strlen else ln
        nr, ng, nc = C.c_int(), C.c_int(), C.c_int()
        rp = R.sdust(None, C.cast(cbuf, C.c_void_p), l_arg, T, W, C.byref(nr))
        ref = np.array([rp[i] for i in range(nr.value)], dtype=np.uint64)
        libc.free(C.cast(rp, C.c_void_p))
        gp = L.cornetto_sdust(None, C.cast(cbuf, C.c_void_p), l_arg, T, W, C.byref(ng))
        ok = ng.value == nr.value and all(gp[i] == ref[i] for i in range(ng.value))
        if gp:
            libc.free(C.cast(gp, C.c_void_p))
        cp = L.cornetto_sdust_core(C.cast(cbuf, C.c_void_p), l_arg, T, W, C.byref(nc), buf)      # (the result belongs to the buf)
        ok = ok and nc.value == nr.value and all(cp[i] == ref[i] for i in range(nc.value))
        words += nr.value
        bases += ln
        if (seed - s0) % 100 == 99:
            print("... %d seeds, %d mismatches so far" % (seed - s0 + 1, bad), flush=True)
        if not ok:
            bad += 1
            print("seed %d: MISMATCH (length %d, T %d, W %d, l_seq %d; reference %d intervals, cornetto_sdust %d, cornetto_sdust_core %d)" % (seed, ln, T, W, l_arg, nr.value, ng.value, nc.value), flush=True)
    L.cornetto_sdust_buf_destroy(buf)
    print("fuzz_abi_sdust: %d seeds from %d (%d bases, %d intervals compared), %d mismatches" % (n, s0, bases, words, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
