"""GPU: the real bench.py control flow with 2 ranks on ONE GPU (the ranks share the device, collectives go over gloo —
bench.py chooses that by itself when there are fewer GPUs than ranks).  The records gathered on rank 0 must be byte-identical
to those of 1-rank runs over the same assemblies:
  * strong scaling: one assembly, contigs split with cornetto_amd.dist.lpt_partition, the 3 x u64 all-reduce behind the
    coverage thresholds, gather in global contig order  ==  the 1-rank run;
  * weak scaling: rank r scans assembly r  ==  the 1-rank runs with --assembly-index 0 and 1.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--gbases", "0.05", "--steps", "2", "--warmup", "1", "--check-steps", "2", "--gather", "--no-cpu", "--no-profiles", "--no-e2e"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(n, extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + COMMON + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + COMMON + extra
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode("replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_two_ranks_strong_scaling_equals_one_rank():
    one = _run(1, ["--scaling", "strong"])
    two = _run(2, ["--scaling", "strong"])
    assert one["scaling"] == "strong" and two["scaling"] == "strong" and two["n_gpus"] == 2
    assert one["determinism"]["identical"] and two["determinism"]["identical"]
    assert len(one["gathered_digests"]) == 1
    assert two["gathered_digests"] == one["gathered_digests"]
    assert two["config"]["bases_job"] == one["config"]["bases_job"]              # one assembly, whatever the rank count
    assert 0 < two["config"]["contigs_rank0"] < one["config"]["contigs_rank0"]


@pytest.mark.timeout(1800)
def test_two_ranks_weak_scaling_equals_two_one_rank_runs():
    a0 = _run(1, ["--assembly-index", "0"])
    a1 = _run(1, ["--assembly-index", "1"])
    two = _run(2, [])
    assert two["scaling"] == "weak" and two["n_gpus"] == 2
    assert a0["gathered_digests"] != a1["gathered_digests"]
    assert two["gathered_digests"] == a0["gathered_digests"] + a1["gathered_digests"]
    assert two["config"]["bases_job"] == 2 * a0["config"]["bases_job"]
