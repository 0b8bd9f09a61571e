"""CPU, world_size 2 over gloo: the multi-GPU plumbing (contig partition, the one all-reduce, the ordered
gather of records to rank 0).  The per-contig compute is replaced by the CPU oracle here — this test is
about the exchange, the GPU kernels are covered by test_gpu_parity.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from cornetto_amd.dist import allreduce_sums, gather_records, lpt_partition  # noqa: E402

IVL_DT = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("finish", "<i4")])


def _contigs():
    rng = np.random.default_rng(42)
    lens = [5000, 120, 0, 900, 15000, 64, 3000, 7000, 1, 2500, 11000]
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seqs = []
    for n in lens:
        s = alpha[rng.integers(0, 4, size=n)].copy()
        if n > 500:
            s[100:400] = np.frombuffer((b"AC" * 150), dtype=np.uint8)
            s[450:460] = ord("N")
        seqs.append(s)
    return seqs


def _local_sdust(seqs, idxs):
    import oracle_bind as ob
    rows = []
    for li, gi in enumerate(idxs):
        for r in ob.sdust(seqs[gi], 20, 64):
            rows.append((li, int(r) >> 32, int(r) & 0xFFFFFFFF))
    return np.array(rows, dtype=IVL_DT) if rows else np.zeros(0, dtype=IVL_DT)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = _contigs()
    parts = lpt_partition([len(s) for s in seqs], world)
    mine = parts[rank]
    recs = _local_sdust(seqs, mine)
    allr = gather_records(recs, mine)
    sums = allreduce_sums((sum(len(seqs[i]) for i in mine), rank + 1, len(mine)))
    if rank == 0:
        q.put((allr.tolist(), sums))
    dist.barrier()
    dist.destroy_process_group()


def test_lpt_partition_is_a_balanced_partition():
    lens = [242_000_000, 200_000_000, 150_000_000, 90_000_000, 61_000_000, 5_000_000, 100_000, 90_000, 0, 7]
    for world in (1, 2, 3, 4, 8):
        parts = lpt_partition(lens, world)
        assert sorted(i for p in parts for i in p) == list(range(len(lens)))
        loads = [sum(lens[i] for i in p) for p in parts]
        assert max(loads) <= sum(lens) / world + max(lens)
        assert all(p == sorted(p) for p in parts)
    assert lpt_partition(lens, 2) == lpt_partition(lens, 2)      # deterministic: no message needed


@pytest.mark.timeout(300)
def test_two_ranks_gather_restores_reference_order():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, sums = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = _contigs()
    exp = _local_sdust(seqs, list(range(len(seqs))))             # single-process order = reference print order
    assert got == exp.tolist()
    assert sums == (sum(len(x) for x in seqs), 3, len(seqs))


def test_single_process_gather_is_identity_order():
    recs = np.array([(1, 5, 9), (0, 1, 2), (1, 20, 30)], dtype=IVL_DT)
    out = gather_records(recs, [7, 3])
    assert out.tolist() == [(3, 5, 9), (3, 20, 30), (7, 1, 2)]
    assert allreduce_sums((1, 2, 3)) == (1, 2, 3)


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cornetto_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_reads_shares_cut_the_stream_at_read_boundaries(world):
    """config 5 (src/seq.c:116-129 per read): the shares of the ranks, in rank order, are the stream; no read is split,
    none is in two shares, and the shares are balanced by bases to within one read"""
    bench = _bench_module()
    rng = np.random.default_rng(7)
    pieces = [np.clip(rng.lognormal(9.2, 0.9, size=n), 200, 200000).astype(np.int64) for n in (700, 650)]
    n_pieces = 5
    shares = bench.reads_shares(pieces, n_pieces, world)
    assert len(shares) == world
    stream = [(j % 2, i) for j in range(n_pieces) for i in range(len(pieces[j % 2]))]
    got = [(s, i) for sh in shares for (s, lo, hi) in sh for i in range(lo, hi)]
    assert got == stream
    B = sum(int(pieces[j % 2].sum()) for j in range(n_pieces))
    per = [sum(int(pieces[s][lo:hi].sum()) for s, lo, hi in sh) for sh in shares]
    assert sum(per) == B
    assert max(abs(p - B / world) for p in per) <= 2 * 200000
    assert all(hi > lo for sh in shares for _, lo, hi in sh)


def test_reads_shares_with_more_ranks_than_reads():
    bench = _bench_module()
    shares = bench.reads_shares([np.array([1000, 10], dtype=np.int64)], 1, 4)
    assert [(s, i) for sh in shares for (s, lo, hi) in sh for i in range(lo, hi)] == [(0, 0), (0, 1)]


def test_plain_bench_with_several_gpus_starts_its_own_ranks_or_refuses():
    """`python bench.py --gpus 2` (the shape of the driver's command) never needs a launcher around it: without two visible
    GPUs it prints no line and exits 3 (here: no GPU at all); the launch itself is covered on the GPU box
    (tests/test_gpu_bench_ranks.py::test_plain_command_starts_its_own_ranks)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the launch is tested by tests/test_gpu_bench_ranks.py")
    assert p.returncode == 3, p.stderr.decode(errors="replace")[-2000:]
    assert p.stdout == b"" and b"refusing" in p.stderr
