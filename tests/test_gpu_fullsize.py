"""GPU, at the bench workload's scale (the synthetic HG002-like assembly of bench.py, 1 Gbp here so that the
test stays under a minute): size-independent properties + oracle spot checks on slices.

  * sdust: the result is canonical (sorted, disjoint, non-adjacent per contig) and identical for different
    chunk sizes / chunk-to-lane mappings (the speculative decomposition must not show in the output);
    small contigs are compared with the oracle in full;
  * telofind: runs are sorted and disjoint per contig and strand, every planted telomere array is found with
    its exact extent, small contigs equal the oracle;
  * coverage windows: device totals equal the exact sums, all windows of small contigs equal the oracle."""
import os
import sys

import numpy as np
import pytest

import oracle_bind as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def world():
    import torch
    import bench
    import cornetto_amd
    dev = torch.device("cuda", 0)
    lens = bench.contig_lengths(1_000_000_000)
    bases, offs = bench.make_assembly(torch, dev, lens, 7)
    depth, mq = bench.make_coverage(torch, dev, lens, offs, 7)
    torch.cuda.synchronize()
    acc = cornetto_amd.Accel(0)
    asm = acc.asm_wrap(bases.data_ptr(), offs, np.array(lens, dtype=np.int64))
    cov = acc.cov_wrap(depth.data_ptr(), mq.data_ptr(), offs, np.array(lens, dtype=np.int32))
    yield dict(torch=torch, acc=acc, asm=asm, cov=cov, lens=lens, offs=offs, bases=bases, depth=depth, mq=mq)
    asm.close()
    cov.close()
    acc.close()


def _small_contigs(lens, limit=400_000, n=6):
    idx = [i for i, x in enumerate(lens) if x <= limit]
    return idx[-n:]


def test_sdust_decomposition_invariance_and_canonical_form(world, monkeypatch):
    acc, asm, lens = world["acc"], world["asm"], world["lens"]
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", "1536")
    a = acc.sdust(asm, 20, 64).copy()
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", "4096")
    b = acc.sdust(asm, 20, 64).copy()
    monkeypatch.setenv("CORNETTO_SDUST_CHUNK", "777")
    monkeypatch.setenv("CORNETTO_SDUST_ORDER", "0")
    monkeypatch.setenv("CORNETTO_SDUST_WAVES", "1000")
    c = acc.sdust(asm, 20, 64).copy()
    assert len(a) > 10000
    assert np.array_equal(a, b) and np.array_equal(a, c)
    # canonical: by contig, start ascending, disjoint and non-adjacent (src/sdust/sdust.c:94-98)
    same = a["ctg"][1:] == a["ctg"][:-1]
    assert np.all(np.diff(a["ctg"]) >= 0)
    assert np.all(a["start"][1:][same] > a["finish"][:-1][same])
    assert np.all(a["finish"] > a["start"])
    # oracle on whole small contigs
    for ci in _small_contigs(lens):
        off = int(world["offs"][ci])
        seq = world["bases"][off:off + lens[ci]].cpu().numpy()
        exp = [(int(r) >> 32, int(r) & 0xFFFFFFFF) for r in ob.sdust(seq, 20, 64)]
        got = [(int(x["start"]), int(x["finish"])) for x in a[a["ctg"] == ci]]
        assert got == exp, ci


def test_telofind_properties_and_planted_arrays(world):
    acc, asm, lens = world["acc"], world["asm"], world["lens"]
    thr = acc.telowin_threshold(0.4, 99.9)
    hits, wins = acc.telo_scan(asm, b"TTAGGG", thr)
    assert len(hits) > 100000
    key = hits["ctg"].astype(np.int64) * 2 + hits["strand"]
    assert np.all(np.diff(key) >= 0)                                    # contig, then strand 0 before strand 1
    same = key[1:] == key[:-1]
    assert np.all(hits["start"][1:][same] > hits["end"][:-1][same])     # disjoint, and never touching (:57 resumes at end+1)
    assert np.all((hits["end"] - hits["start"]) % 6 == 0)
    for ci, n in enumerate(lens):
        if n < 50000:
            continue
        h = hits[hits["ctg"] == ci]
        rev = h[h["strand"] == 1]
        fwd = h[h["strand"] == 0]
        assert (0, 12000) in {(int(x["start"]), int(x["end"])) for x in rev}, ci          # CCCTAA x 2000 at the start
        assert (n - 9000, n) in {(int(x["start"]), int(x["end"])) for x in fwd}, ci       # TTAGGG x 1500 at the end
        w = wins[wins["ctg"] == ci]
        assert len(w) > 0 and int(w["start"][0]) == 0
    for ci in _small_contigs(lens):
        off = int(world["offs"][ci])
        seq = world["bases"][off:off + lens[ci]].cpu().numpy()
        oh = ob.telofind(seq, b"TTAGGG")
        exp = [(int(x["strand"]), int(x["start"]), int(x["end"])) for x in oh]
        got = [(int(x["strand"]), int(x["start"]), int(x["end"])) for x in hits[hits["ctg"] == ci]]
        assert got == exp, ci
        ew = [(int(x["start"]), int(x["end"]), int(x["car"])) for x in ob.telowin(oh, lens[ci], thr)]
        gw = [(int(x["start"]), int(x["end"]), int(x["car"])) for x in wins[wins["ctg"] == ci]]
        assert gw == ew, ci


def test_coverage_totals_and_windows(world):
    acc, cov, lens, torch = world["acc"], world["cov"], world["lens"], world["torch"]
    sd, sq, n = acc.cov_prepare(cov, 2500, 50)
    tot_d = tot_q = 0
    for off, ln in zip(world["offs"], lens):
        tot_d += int(world["depth"][int(off):int(off) + ln].to(torch.int64).sum().item())
        tot_q += int(world["mq"][int(off):int(off) + ln].to(torch.int64).sum().item())
    assert (sd, sq, n) == (tot_d, tot_q, sum(lens))
    for ci in _small_contigs(lens):
        off = int(world["offs"][ci])
        d = world["depth"][off:off + lens[ci]].cpu().numpy().view(np.uint16)
        q = world["mq"][off:off + lens[ci]].cpu().numpy().view(np.uint16)
        got = acc.cov_regs(cov, ci)
        exp = ob.get_regs(d, q, 2500, 50)
        assert np.array_equal(got, exp.astype(got.dtype)), ci
    mean = int(np.floor(sd / n + 0.5))
    lo, hi = acc.cov_threshold(0.4, mean), acc.cov_threshold(2.5, mean)
    recs = acc.cov_select(cov, lo, hi, 0.4, 100000, 1000000, False)
    assert len(recs) > 100000
    key = recs["ctg"].astype(np.int64) * (1 << 32) + recs["st"]
    assert np.all(np.diff(key) > 0)                                      # print order: contig, then window
    assert np.all(recs["st"] % 50 == 0)
    fun = (recs["depth"] < lo) | (recs["depth"] > hi) | (recs["mq_depth"] / np.maximum(recs["depth"], 1e-300) < np.float32(0.4))
    assert np.all(fun | (recs["depth"] == 0))
