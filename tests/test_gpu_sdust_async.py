"""cornetto_sdust_asm_begin() / _end(): the call of cornetto_sdust_asm() in two parts (include/cornetto_accel.h) — same intervals whatever _begin could
queue: nothing (first call for an assembly), the whole call in one go (a repeated call), one go with estimates that do not hold (the long way from
_end), another assembly in between, a _begin nobody finished."""
import os

import numpy as np
import pytest

import cornetto_amd

pytestmark = pytest.mark.gpu


def _asm(acc, seed, n=6, lo=3000, hi=400000):
    """contigs with what sdust finds: homopolymer and short-unit runs, an N run, lower case, on random sequence"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = []
    for ln in [int(x) for x in rng.integers(lo, hi, size=n)] + [70, 3]:
        s = acgt[rng.integers(0, 4, size=ln)].copy()
        for _ in range(max(1, ln // 700)):
            p = int(rng.integers(0, max(1, ln - 400)))
            unit = [b"A", b"AT", b"CAG", b"TTAGGG", b"N", b"acgt"][int(rng.integers(0, 6))]
            rep = np.frombuffer(unit * int(rng.integers(8, 120)), dtype=np.uint8)
            s[p:p + len(rep)] = rep[:len(s[p:p + len(rep)])]
        seqs.append(s)
    return acc.asm_upload(seqs), seqs


@pytest.fixture(scope="module")
def acc():
    a = cornetto_amd.Accel(0)
    yield a
    a.close()


def test_begin_end_equal_the_plain_call(acc):
    asm, _ = _asm(acc, 11)
    try:
        ref = acc.sdust(asm, 20, 64).copy()
        for rep in range(3):                       # (the first pair: nothing to size the call by or estimates from the plain call; then one go)
            before = acc.launch_count()
            acc.sdust_begin(asm, 20, 64)
            queued = acc.launch_count() != before
            got = acc.sdust_end(asm, 20, 64)
            assert got.tobytes() == ref.tobytes(), rep
            if rep:
                assert queued                       # the counts of the call before size this one: queued by _begin
        # other parameters: their own estimates
        ref2 = acc.sdust(asm, 25, 40).copy()
        acc.sdust_begin(asm, 25, 40)
        assert acc.sdust_end(asm, 25, 40).tobytes() == ref2.tobytes()
    finally:
        asm.close()


def test_first_call_queues_nothing_and_end_runs_it(acc):
    asm, _ = _asm(acc, 12)
    try:
        before = acc.launch_count()
        acc.sdust_begin(asm, 20, 64)                # no earlier call: nothing queued
        assert acc.launch_count() == before
        got = acc.sdust_end(asm, 20, 64).copy()
        assert got.tobytes() == acc.sdust(asm, 20, 64).tobytes()
    finally:
        asm.close()


def test_estimates_that_do_not_hold_take_the_long_way(dacc):
    acc = dacc                                   # the development build: the switches below exist there only
    asm, _ = _asm(acc, 13, n=4, lo=200000, hi=900000)
    try:
        ref = acc.sdust(asm, 20, 64).copy()
        assert len(ref) > 64
        os.environ["CORNETTO_SDUST_EST_FORCE"] = "16"      # room for 16 rows: the stitch refuses, _end repeats the call
        try:
            acc.sdust_begin(asm, 20, 64)
            got = acc.sdust_end(asm, 20, 64)
        finally:
            del os.environ["CORNETTO_SDUST_EST_FORCE"]
        assert got.tobytes() == ref.tobytes()
        acc.sdust_begin(asm, 20, 64)                        # and the next pair is in one go again
        assert acc.sdust_end(asm, 20, 64).tobytes() == ref.tobytes()
    finally:
        asm.close()


def test_end_for_another_call_and_a_begin_left_alone(acc):
    a1, _ = _asm(acc, 14)
    a2, _ = _asm(acc, 15)
    try:
        r1 = acc.sdust(a1, 20, 64).copy()
        r2 = acc.sdust(a2, 20, 64).copy()
        acc.sdust_begin(a1, 20, 64)
        assert acc.sdust_end(a2, 20, 64).tobytes() == r2.tobytes()      # not what was begun: that one is dropped, this one computed
        acc.sdust_begin(a1, 20, 64)
        assert acc.sdust(a2, 20, 64).tobytes() == r2.tobytes()          # a plain call finishes and drops a pending _begin
        assert acc.sdust(a1, 20, 64).tobytes() == r1.tobytes()
        acc.sdust_begin(a1, 20, 64)
        with pytest.raises(Exception):
            acc.sdust_begin(a1, 20, 64)                                 # twice without _end
        assert acc.sdust_end(a1, 20, 64).tobytes() == r1.tobytes()
    finally:
        a1.close()
        a2.close()


def test_close_with_a_call_begun():
    a = cornetto_amd.Accel(0)
    asm, _ = _asm(a, 16)
    a.sdust(asm, 20, 64)
    a.sdust_begin(asm, 20, 64)                  # queued, never finished
    asm.close()
    a.close()                                   # waits for the stream, gives the result array back
