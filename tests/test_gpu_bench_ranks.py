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
COMMON = ["--gbases", "0.05", "--steps", "2", "--warmup", "1", "--check-steps", "2", "--gather", "--no-cpu", "--no-profiles", "--no-e2e", "--no-reads",
          "--emulate-ranks", "", "--no-second"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _n_gpus():
    import torch
    return torch.cuda.device_count()


def _run(n, extra, rc_ok=(0,)):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + COMMON + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + COMMON + extra
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd=ROOT)
    assert p.returncode in rc_ok, p.stderr.decode(errors="replace")[-3000:]
    if p.returncode != 0:
        return {"rc": p.returncode, "stderr": p.stderr.decode(errors="replace")}
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_two_ranks_strong_scaling_equals_one_rank():
    shared = [] if _n_gpus() >= 2 else ["--allow-shared-device"]
    one = _run(1, ["--scaling", "strong"])
    two = _run(2, ["--scaling", "strong"] + shared)
    assert one["scaling"] == "single" and two["scaling"] == "strong" and two["n_gpus"] == 2      # (one GPU has no scaling mode to name)
    assert one["determinism"]["identical"] and two["determinism"]["identical"]
    assert len(one["gathered_digests"]) == 1
    assert two["gathered_digests"] == one["gathered_digests"]
    assert two["config"]["bases_job"] == one["config"]["bases_job"]              # one assembly, whatever the rank count
    assert 0 < two["config"]["contigs_rank0"] < one["config"]["contigs_rank0"]
    c = two["collectives"]
    assert c["world"] == 2 and len(c["devices"]) == 2 and len(c["per_rank_ms"]) == 2 and c["allreduce_3xi64_us"] > 0
    assert c["backend"] == ("nccl" if _n_gpus() >= 2 else "gloo") and c["distinct_devices"] == (_n_gpus() >= 2)


@pytest.mark.timeout(1800)
def test_two_ranks_over_pieces_of_cut_contigs_equal_one_rank():
    """--split-tol -1: the ranks take [r, r + 1) x half of the bases, the contig the border falls into is cut on a clean position and both ranks
    scan their piece of it with halos (cornetto_amd.dist.SplitPlan) — the records rank 0 puts together are those of the one-rank run"""
    shared = [] if _n_gpus() >= 2 else ["--allow-shared-device"]
    big = ["--gbases", "0.2"]                      # (200 Mb: half of the bases ends 7.6 Mb into a contig of 9.3 Mb; at 50 Mb no contig is long enough to be cut)
    one = _run(1, ["--scaling", "strong"] + big)
    two = _run(2, ["--scaling", "strong", "--split-tol", "-1"] + big + shared)
    assert two["config"].get("cut_contigs", 0) >= 1
    assert one["determinism"]["identical"] and two["determinism"]["identical"]
    assert two["gathered_digests"] == one["gathered_digests"]
    assert two["config"]["bases_job"] == one["config"]["bases_job"]


@pytest.mark.timeout(900)
def test_ranks_sharing_a_device_are_refused():
    """a --gpus 2 run whose ranks land on ONE device must not print a multi-GPU line (exit 3) unless the test flag is given"""
    if _n_gpus() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    r = _run(2, ["--scaling", "strong"], rc_ok=(1, 3))       # torch.distributed.run reports the failure of its workers as 1
    assert "share one device" in r["stderr"]


@pytest.mark.timeout(1800)
def test_default_of_two_ranks_is_strong_with_a_second_weak_measurement():
    """`--gpus 2` without --scaling: the metric's mode (ONE assembly, strong scaling) is `value`, the other mode with the gather of
    all records comes second in the same line"""
    shared = [] if _n_gpus() >= 2 else ["--allow-shared-device"]
    env_common = [x for x in COMMON if x not in ("--no-second", "--gather")]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + env_common + shared
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
    assert line["scaling"] == "strong" and line["second"]["scaling"] == "weak" and line["second"]["gather"] is True
    assert line["second"]["bases_job"] == 2 * line["config"]["bases_job"] and line["second"]["value"] > 0


@pytest.mark.timeout(900)
def test_scaling_model_of_one_rank():
    one = _run(1, ["--emulate-ranks", "2,4"])
    m = one["scaling_model"]
    assert m["modelled"] is True and set(m) >= {"2", "4"}
    for n in ("2", "4"):
        assert len(m[n]["per_rank_ms"]) == int(n) and sum(m[n]["bases_per_rank"]) == one["config"]["bases_job"]
        assert m[n]["step_ms"] == max(m[n]["per_rank_ms"]) and 0 < m[n]["efficiency"] <= 1.5


@pytest.mark.timeout(1800)
def test_two_ranks_weak_scaling_equals_two_one_rank_runs():
    shared = [] if _n_gpus() >= 2 else ["--allow-shared-device"]
    a0 = _run(1, ["--assembly-index", "0"])
    a1 = _run(1, ["--assembly-index", "1"])
    two = _run(2, ["--scaling", "weak"] + shared)
    assert two["scaling"] == "weak" and two["n_gpus"] == 2
    assert a0["gathered_digests"] != a1["gathered_digests"]
    assert two["gathered_digests"] == a0["gathered_digests"] + a1["gathered_digests"]
    assert two["config"]["bases_job"] == 2 * a0["config"]["bases_job"]


@pytest.mark.timeout(1800)
def test_plain_command_starts_its_own_ranks():
    """the shape of the command the driver runs — `python bench.py --gpus N ...`, no torch.distributed.run around it: bench.py starts
    the ranks itself as child processes and relays ONE line: config 4 (`second`: the other scaling mode with the gather) and config 5
    (`reads`: one FASTQ stream sharded over the ranks by cumulative bases) are in it"""
    shared = [] if _n_gpus() >= 2 else ["--allow-shared-device"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--gbases", "0.05", "--steps", "2", "--warmup", "1", "--check-steps", "2",
           "--reads-gbases", "0.08"] + shared
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout.decode()[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["collectives"]["world"] == 2
    assert line["second"]["scaling"] == "weak" and line["second"]["gather"] is True
    rd = line["reads"]
    assert rd["ranks"] == 2 and rd["shares_add_up"] is True and len(rd["per_rank"]) == 2
    assert sum(x["bases_in"] for x in rd["per_rank"]) == rd["bases_in"] and rd["sdust_intervals"] > 0
    assert abs(rd["per_rank"][0]["bases_in"] - rd["per_rank"][1]["bases_in"]) <= 400000
    assert rd.get("parity", {"ok": True})["ok"]


@pytest.mark.timeout(600)
def test_plain_command_refuses_shared_devices():
    if _n_gpus() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=500, cwd=ROOT)
    assert p.returncode == 3 and p.stdout == b"" and b"refusing" in p.stderr
