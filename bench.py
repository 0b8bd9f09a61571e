#!/usr/bin/env python3
"""bench.py — Gbases/s scanned by the panel-creation hot path (telofind+telowin, sdust, (no)boringbits
window stage) on a synthetic ~3 Gbp HG002-like assembly (BASELINE.json metric; SURVEY 8d inputs).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W [--scaling strong] [--gather]

One process per GPU.  A "step" is one pass of the hot path over the rank's contigs, inputs resident in HBM
(bases 1 B/base; depth + mq 2 x u16/base), results (telomere runs, telomere windows, sdust intervals, selected
coverage windows) delivered to the host memory of the rank that owns the contigs (where a sharded run writes its
part of the BED/TSV output; `--gather` additionally collects every record on rank 0).

  --scaling weak    (default) every rank holds its own assembly (config 4 of BASELINE.json: N iteration assemblies)
  --scaling strong  ONE assembly; its contigs are split over the ranks by cornetto_amd.dist.lpt_partition
In both, the only data-path collective is the all-reduce of the 3 x u64 depth totals behind the coverage thresholds.

Rank 0 prints ONE JSON line.  Beside the contract's fields it carries
  roofline        the dominant kernel (sdust_kernel) from HIP events on its launch stream
  cpu_baseline    the reference's own functions (oracle/_ref, "reference") or the CPU oracle ("port") on 1 host core
                  over the leading contigs of the same workload (N=1 only)
  parity          the GPU results of the timed steps compared, record for record, with what that CPU leg computed
                  for the same contigs (the process exits 1 on a mismatch)
  determinism     digest of the four result arrays of every step of an extra, untimed run of --check-steps steps
  profiles        N=1: the same step on a satellite-dense assembly (--profile satellite makes that the main workload)
  e2e             N=1: wall time of the C CLI on the same assembly written as a FASTA file
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
PMC_ROUNDS = ("r06", "r05", "r04")                                         # the newest committed counter passes first
PMC_FILE = os.path.join("profiles", "%s_%s_pmc_traffic_%s.json")    # % (round, workload profile, kernel symbol)
SQ_FILE = os.path.join("profiles", "%s_%s_sq_%s.json")


from cornetto_amd.synth import contig_lengths, make_assembly, make_coverage, make_bedgraph_text, make_fastq_piece, FQ_HEAD   # noqa: E402,F401  (tests and tools import the generators from cornetto_amd.synth)


def cornetto_amd_dt(name):
    import cornetto_amd
    return getattr(cornetto_amd, name)


REF_SO = os.path.join(ROOT, "oracle", "_ref", "libcornetto_ref.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "cornetto")
HIT_KEYS = ("strand", "start", "end")


class _RefLib:
    """The reference's own per-contig functions out of oracle/_ref/libcornetto_ref.so (built from the sources under
    /root/reference by oracle/ref.mk; git-ignored, travels with the snapshot).  Timed as they are: `find` and
    `process_scaffold` print their records themselves (find_telomere.c:44, telomere_windows.c:28), so stdout points at
    a file in memory (/dev/shm) while they run and the parity check reads the records back from it; `sdust` is
    sdust.c:162; `get_regs` is boringbits_main.c:322 over the structures of boringbits_main.c:116-147."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.so = C.CDLL(REF_SO)
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]
        self.so.find.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        self.so.find.restype = None
        self.so.process_scaffold.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
        self.so.process_scaffold.restype = None
        self.so.sdust.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        self.so.sdust.restype = C.c_void_p

        class CtgDepth(C.Structure):
            _fields_ = [("ctg_name", C.c_char_p), ("ctg_length", C.c_int), ("c_depth", C.c_int),
                        ("depth", C.c_void_p), ("mq_depth", C.c_void_p)]

        class AsmDepth(C.Structure):
            _fields_ = [("num_ctg", C.c_int), ("c_ctg", C.c_int), ("ctg_depth", C.POINTER(CtgDepth)),
                        ("mean_depth", C.c_int), ("mean_mq_depth", C.c_int)]

        class CtgReg(C.Structure):
            _fields_ = [("ctg_name", C.c_char_p), ("ctg_length", C.c_int), ("n_reg", C.c_int), ("reg", C.c_void_p)]

        class AsmReg(C.Structure):
            _fields_ = [("num_ctg", C.c_int), ("ctg_reg", C.POINTER(CtgReg)), ("mean_depth", C.c_int), ("mean_mq_depth", C.c_int)]
        self.CtgDepth, self.AsmDepth, self.AsmReg = CtgDepth, AsmDepth, AsmReg
        self.so.get_regs.argtypes = [C.POINTER(AsmDepth), C.c_int, C.c_int]
        self.so.get_regs.restype = C.POINTER(AsmReg)
        self.so.free_asm_reg.argtypes = [C.POINTER(AsmReg)]
        self.so.free_asm_reg.restype = None
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
        self.capfile = os.path.join(shm, "cornetto_bench_ref_stdout.%d" % os.getpid())

    def captured(self, fn, *a):
        """run fn(*a) with file descriptor 1 on a file in memory (the reference functions printf their records);
        -> (seconds, the bytes printed)"""
        sys.stdout.flush()
        keep = os.dup(1)
        fd = os.open(self.capfile, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        try:
            os.dup2(fd, 1)
            t0 = time.perf_counter()
            fn(*a)
            self.libc.fflush(None)
            dt = time.perf_counter() - t0
        finally:
            os.dup2(keep, 1)
            os.close(keep)
            os.close(fd)
        with open(self.capfile, "rb") as f:
            out = f.read()
        os.remove(self.capfile)
        return dt, out


def _sample_contigs(lens, budget_bases):
    """the leading WHOLE contigs within the budget (at least one; a first contig beyond the budget is cut) — in the bench assembly the
    two largest — plus three small ones that cost nothing and bring the other predicates into the in-run parity: the smallest contig of
    at least 1 Mb (the -m boundary of print_fun_bits, windows against the edges), one of about 10 Mb, and the smallest contig of all
    (below -m: no window is selected, boringbits_main.c:428)"""
    out, done = [], 0
    for i, n in enumerate(lens):
        if out and done + n > budget_bases:
            break
        n = int(min(n, budget_bases)) if not out else int(n)
        out.append((i, n))
        done += n
    have = {i for i, _ in out}
    if out and out[0][1] == lens[out[0][0]]:
        big = [(n, i) for i, n in enumerate(lens) if n >= 1_000_000 and i not in have]
        mid = [(abs(n - 10_000_000), i) for i, n in enumerate(lens) if i not in have and 1_000_000 <= n <= 40_000_000]
        small = [(n, i) for i, n in enumerate(lens) if i not in have]
        extra = []
        if big:
            extra.append(min(big)[1])
        if mid:
            extra.append(min(mid)[1])
        if small:
            extra.append(min(small)[1])
        for i in dict.fromkeys(extra):
            out.append((i, int(lens[i])))
    return out


def cpu_all_cores(R, per_contig, lens_all):
    """SURVEY 8d's second CPU line: the reference takes -t and ignores it (src/boringbits_main.c:590-595), so "all host cores" can only
    mean one process per contig.  Measured: the sample's contigs at the same time, one thread each (ctypes drops the GIL; the reference's
    functions share nothing), against the one-after-the-other time of the same contigs -> how well the host scales per contig.  The
    whole-assembly figure is that per-core rate over min(cores, contigs) cores, BOUNDED by the largest contig, which one core has to
    walk alone: bases / max(t_largest, sum_t / cores)."""
    import threading
    C = R.C
    jobs = [j for j in per_contig if j["len"] >= 1_000_000][:max(2, min(8, os.cpu_count() or 2))]
    if len(jobs) < 2:
        return None

    def work(j):
        cnt = C.c_int()
        r = R.so.sdust(None, j["seq"].ctypes.data, j["len"], 20, 64, C.byref(cnt))
        R.libc.free(r)
        ctg = R.CtgDepth(b"c", j["len"], j["len"], j["d"].ctypes.data, j["q"].ctypes.data)
        asm = R.AsmDepth(1, 1, C.pointer(ctg), 30, 30)
        R.so.free_asm_reg(R.so.get_regs(C.byref(asm), 2500, 50))
    th = [threading.Thread(target=work, args=(j,)) for j in jobs]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    serial = sum(j["t_sdust_regs"] for j in jobs)
    bases = sum(j["len"] for j in jobs)
    cores = os.cpu_count() or 1
    rate1 = sum(j["len"] for j in per_contig) / sum(j["t_all"] for j in per_contig)          # bases per second and core, all four stages
    slow = wall / max(j["t_sdust_regs"] for j in jobs)                                        # >1: the contigs slow each other down (memory bandwidth)
    total = float(sum(lens_all))
    t_largest = max(lens_all) / rate1 * max(1.0, slow)
    t_spread = total / rate1 / min(cores, len(lens_all)) * max(1.0, slow)
    return {"value": round(total / max(t_largest, t_spread) / 1e9, 4), "unit": "Gbases/s", "cores": min(cores, len(lens_all)), "modelled": True,
            "measured": {"threads": len(jobs), "bases": bases, "wall_s": round(wall, 2), "one_after_the_other_s": round(serial, 2),
                         "stages": "sdust + get_regs (find / process_scaffold print to the one stdout: not run side by side)",
                         "slowdown_of_the_slowest_contig_beside_the_others": round(slow, 3)},
            "bound": "one process per contig (the reference has no threads: -t is accepted and ignored, src/boringbits_main.c:590-595): the largest contig "
                     "(%d bases) takes %.1f s on one core, the other %d contigs fit beside it on %d cores -> %.3f Gbases/s for the assembly however many cores the host has" % (
                         max(lens_all), t_largest, len(lens_all) - 1, cores, total / max(t_largest, t_spread) / 1e9)}


def cpu_reference_leg(bases, depth, mq, offs, lens, own, budget_bases, pick=None):
    """The CPU side of the run on one host core: the reference itself (oracle/_ref) where it was built, else the oracle
    port, over the leading contigs `own[0..]` of the same workload (or over the local contigs `pick` names, whole).  Returns
    (cpu_baseline dict, per-contig results for the parity check)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bind as ob
    use_ref = os.path.exists(REF_SO) and os.environ.get("CORNETTO_BENCH_BASELINE", "reference") != "port"
    R = _RefLib() if use_ref else None
    if not use_ref:
        ob.lib()
    thr = ob.telowin_threshold(0.4, 99.9)
    t = {"telofind": 0.0, "telowin": 0.0, "sdust": 0.0, "get_regs": 0.0}
    results = []
    per_contig = []
    sample = _sample_contigs([lens[i] for i in own], budget_bases) if pick is None else [(li, int(lens[own[li]])) for li in pick]
    done = 0
    for li, n in sample:
        gi = own[li]
        off = int(offs[gi])
        seq = np.ascontiguousarray(np.concatenate([bases[off:off + n].cpu().numpy(), np.zeros(1, np.uint8)]))
        d = np.ascontiguousarray(depth[off:off + n].cpu().numpy().view(np.uint16))
        q = np.ascontiguousarray(mq[off:off + n].cpu().numpy().view(np.uint16))
        res = {"local": li, "len": n, "whole": n == lens[gi]}
        if use_ref:
            C = R.C
            up = seq.copy()
            up[:n] = np.where((up[:n] >= 97) & (up[:n] <= 122), up[:n] - 32, up[:n])   # find_telomere.c:76-81 upper-cases the contig first (disambiguate)
            t0 = time.perf_counter()
            dt, txt = R.captured(R.so.find, up.ctypes.data, b"c", b"TTAGGG")      # find(query = the contig, name, target = the motif)
            t["telofind"] += dt
            rows = np.array([[int(x) for x in l.split(b"\t")[2:5]] for l in txt.splitlines()], dtype=np.int64).reshape(-1, 3)
            res["hits"] = rows                                             # strand, start, end in the reference's print order
            t0 = time.perf_counter()
            marks = np.zeros(n, np.uint8)                                  # telomere_windows.c:69-79: calloc + mark
            for st, en in zip(rows[:, 1].tolist(), rows[:, 2].tolist()):
                marks[st:en] = 1
            t["telowin"] += time.perf_counter() - t0
            # called directly, process_scaffold compares with its file-static THRESHOLD = 0.4 (telomere_windows.c:19,36;
            # telomere_windows_main would lower it to 0.4 * 0.999^6, :53-54): the parity check filters the GPU windows to that
            dt, txt = R.captured(R.so.process_scaffold, b"c", marks.ctypes.data, n)
            t["telowin"] += dt
            res["wins_text"] = txt
            res["wins_thr"] = 0.4
            cnt = C.c_int()
            t_before = dict(t)
            t0 = time.perf_counter()
            r = R.so.sdust(None, seq.ctypes.data, n, 20, 64, C.byref(cnt))
            t["sdust"] += time.perf_counter() - t0
            if r and cnt.value > 0:
                res["sdust"] = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_uint64)), shape=(cnt.value,)).copy()
            else:                                                          # sdust.c:194-198: nothing found = no buffer at all
                res["sdust"] = np.zeros(0, np.uint64)
            R.libc.free(r)
            ctg = R.CtgDepth(b"c", n, n, d.ctypes.data, q.ctypes.data)
            asm = R.AsmDepth(1, 1, C.pointer(ctg), 30, 30)
            t0 = time.perf_counter()
            regs = R.so.get_regs(C.byref(asm), 2500, 50)
            t["get_regs"] += time.perf_counter() - t0
            cr = regs.contents.ctg_reg[0]
            res["regs"] = np.ctypeslib.as_array(C.cast(cr.reg, C.POINTER(C.c_int32)), shape=(cr.n_reg, 4)).copy()
            R.so.free_asm_reg(regs)
            per_contig.append({"len": n, "seq": seq, "d": d, "q": q, "t_sdust_regs": (t["sdust"] - t_before["sdust"]) + (t["get_regs"] - t_before["get_regs"]),
                               "t_all": 0.0})
        else:
            t0 = time.perf_counter()
            hits = ob.telofind(seq[:n], b"TTAGGG")
            t1 = time.perf_counter()
            wins = ob.telowin(hits, n, thr)
            t2 = time.perf_counter()
            res["sdust"] = np.asarray(ob.sdust(seq[:n], 20, 64), dtype=np.uint64)
            t3 = time.perf_counter()
            regs = ob.get_regs(d, q, 2500, 50)
            t4 = time.perf_counter()
            t["telofind"] += t1 - t0
            t["telowin"] += t2 - t1
            t["sdust"] += t3 - t2
            t["get_regs"] += t4 - t3
            res["hits"] = np.stack([hits[k].astype(np.int64) for k in HIT_KEYS], axis=1).reshape(-1, 3)
            res["wins_text"] = b"".join(b"Window\tc\t%d\t%d\t%d\t%s\n" % (n, w["start"], w["end"], ("%.3g" % (float(w["car"]) / float(w["end"] - w["start"]))).encode()) for w in wins)
            res["wins_thr"] = thr
            res["regs"] = np.stack([regs[k].astype(np.int32) for k in ("st", "end", "depth", "mq_depth")], axis=1)
        results.append(res)
        done += n
        if per_contig and per_contig[-1]["t_all"] == 0.0:
            per_contig[-1]["t_all"] = sum(t.values()) - sum(pc["t_all"] for pc in per_contig[:-1])
    total = sum(t.values())
    model = "unknown"
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    what = ("the reference's own find, process_scaffold, sdust(T=20,W=64) and get_regs(2500,50) (oracle/_ref/libcornetto_ref.so, gcc -O2)"
            if use_ref else "the oracle port (oracle/oracle.c, gcc -O2): telofind, telowin, sdust -w64 -t20, get_regs(2500,50)")
    base = {
        "value": round(done / total / 1e9, 6), "unit": "Gbases/s", "cores": 1, "kind": "reference" if use_ref else "port",
        "host_cpu": model, "host_cores": os.cpu_count(),
        "sample": "the %d leading whole contigs (%d bases) of the same synthetic assembly and coverage through %s, each over all "
                  "of it, one thread; %.1f s of CPU" % (len(sample), done, what, total),
        "stage_gbases_s": {k: round(done / v / 1e9, 4) for k, v in t.items()},
    }
    if use_ref and pick is None and os.environ.get("CORNETTO_BENCH_ALL_CORES", "1") != "0":
        try:
            ac = cpu_all_cores(R, per_contig, [lens[i] for i in own])
            if ac:
                base["all_cores"] = ac
        except Exception as e:                      # (a reported extra: never the reason a bench line is lost)
            base["all_cores"] = {"error": repr(e)}
    return base, results


def check_parity(results, gpu, lo, hi, low_mq, min_ctg_len, lens_own):
    """GPU records of one step (hits, wins, ivls, recs: structured arrays with a local "ctg" index) against the CPU
    leg's per-contig results.  -> parity dict ("ok" False and a "first_mismatch" text on any difference)"""
    hits, wins, ivls, recs = gpu
    out = {"ok": True, "contigs": 0, "checked_bases": 0, "telofind_hits": 0, "telowin_windows": 0, "sdust_intervals": 0,
           "cov_windows_all": 0, "cov_windows_selected": 0}

    def bounds(arr, li):
        return np.searchsorted(arr["ctg"], li, "left"), np.searchsorted(arr["ctg"], li, "right")

    def fail(msg):
        if out["ok"]:
            out["ok"] = False
            out["first_mismatch"] = msg

    for r in results:
        li, n = r["local"], r["len"]
        if not r["whole"]:
            continue                                   # a cut contig ends differently from the GPU's whole one
        a, b = bounds(hits, li)
        g = np.stack([hits[k][a:b].astype(np.int64) for k in HIT_KEYS], axis=1).reshape(-1, 3)
        if g.shape != r["hits"].shape or not np.array_equal(g, r["hits"]):
            fail("telofind, local contig %d: %d GPU runs vs %d" % (li, len(g), len(r["hits"])))
        out["telofind_hits"] += len(g)
        a, b = bounds(wins, li)
        txt = b"".join(b"Window\tc\t%d\t%d\t%d\t%s\n" % (n, w["start"], w["end"], ("%.3g" % (float(w["car"]) / float(w["end"] - w["start"]))).encode())
                       for w in wins[a:b] if float(w["car"]) / float(w["end"] - w["start"]) >= r["wins_thr"])
        if txt != r["wins_text"]:
            fail("telowin, local contig %d: %d GPU windows vs %d lines" % (li, b - a, r["wins_text"].count(b"\n")))
        out["telowin_windows"] += txt.count(b"\n")
        a, b = bounds(ivls, li)
        g = (ivls["start"][a:b].astype(np.uint64) << np.uint64(32)) | ivls["finish"][a:b].astype(np.uint32).astype(np.uint64)
        if len(g) != len(r["sdust"]) or not np.array_equal(g, r["sdust"]):
            fail("sdust, local contig %d: %d GPU intervals vs %d" % (li, len(g), len(r["sdust"])))
        out["sdust_intervals"] += len(g)
        # print_fun_bits (boringbits_main.c:425-445) over the reference's own window table
        regs = r["regs"]
        dep, mqd = regs[:, 2].astype(np.int64), regs[:, 3].astype(np.int64)
        with np.errstate(divide="ignore", invalid="ignore"):
            flag = (dep < lo) | (dep > hi) | ((mqd.astype(np.float64) / dep.astype(np.float64)) < np.float64(np.float32(low_mq)))
        exp = regs[flag] if lens_own[li] >= min_ctg_len else regs[:0]
        a, b = bounds(recs, li)
        g = np.stack([recs[k][a:b].astype(np.int32) for k in ("st", "end", "depth", "mq_depth")], axis=1).reshape(-1, 4)
        if g.shape != exp.shape or not np.array_equal(g, exp):
            fail("coverage windows, local contig %d: %d GPU rows vs %d" % (li, len(g), len(exp)))
        out["cov_windows_all"] += len(regs)
        out["cov_windows_selected"] += len(g)
        out["contigs"] += 1
        out["checked_bases"] += n
    return out


def digest(arrs):
    """one digest over the raw bytes of result arrays"""
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except Exception:
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    for a in arrs:
        a = np.ascontiguousarray(a)
        h.update(np.int64(len(a)).tobytes())
        h.update(memoryview(a.view(np.uint8).reshape(-1)))
    return h.hexdigest()


class _Handoff:
    """one value from one thread to another: put() never waits, get() blocks in the lock's own wait (no GIL, no condition variable) —
    the two hand-overs of a step cost a few microseconds instead of the tens a queue.Queue takes, which a 1.5 ms step of a 395 Mb share notices"""

    def __init__(self):
        import _thread
        self._lock = _thread.allocate_lock()
        self._lock.acquire()
        self._value = None

    def put(self, value):
        self._value = value
        self._lock.release()

    def get(self, spin_s=0.0):
        """spin_s > 0: poll for that long first (time.sleep(0) drops the GIL every turn) — the waiter of the hand-over that starts a step is
        awake when the value arrives instead of being woken (~30 us of a 1.2 ms step)"""
        if spin_s > 0:
            t_end = time.perf_counter() + spin_s
            while time.perf_counter() < t_end:
                if self._lock.acquire(False):
                    return self._value
                time.sleep(0)
        self._lock.acquire()
        return self._value


class Rank:
    """one process = one GPU: handles, streams and the resident workload of this rank"""

    def __init__(self, args, torch, dist, cornetto_amd):
        self.args, self.torch, self.dist = args, torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, self.world, args.gpus))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (there is no CPU fallback of the product path)")
        ndev = torch.cuda.device_count()
        self.local_dev = local % max(1, ndev)
        torch.cuda.set_device(self.local_dev)
        self.dev = torch.device("cuda", self.local_dev)
        self.cdev = self.dev                         # device of the tensors handed to collectives
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if ndev >= self.world:
                dist.init_process_group("nccl", device_id=self.dev)      # RCCL over xGMI, one GPU per rank
            else:
                # fewer GPUs than ranks (a 1-GPU test box): ranks share devices and the collectives go over gloo;
                # exercises the same multi-process control flow, not a performance configuration
                dist.init_process_group("gloo")
                self.cdev = torch.device("cpu")
        # the short HBM-bound kernels (telofind, coverage) go on a HIGH-priority stream so that they are not
        # starved by the long sdust kernel of the second stream, which fills every wave slot of the chip
        self.stream = torch.cuda.Stream(device=self.dev, priority=-1)
        self.acc = cornetto_amd.Accel(self.local_dev, self.stream.cuda_stream)
        self.acc2 = cornetto_amd.Accel(self.local_dev, None)               # second stream, same device: the sdust side
        # every entry point once on a built-in 4 kb input (cornetto_accel_warm): the runtime's first-use costs — code objects, copy engines, the first
        # pinned pools — belong to opening the device, not to the first assembly (the CLI does the same behind its open, beside reading the file)
        t_w = time.perf_counter()
        self.lazy = not args.serial and os.environ.get("CORNETTO_BENCH_LAZY", "1") != "0"
        self.acc.set_lazy(self.lazy)                 # (in front of the warm-up: the copy stream exists when its copy queues are set up)
        if os.environ.get("CORNETTO_BENCH_WARM", "1") != "0":
            self.acc.warm(6)
            self.acc2.warm(1)
        self.warm_ms = (time.perf_counter() - t_w) * 1e3
        self.acc.set_timing(args.timing)
        self.acc2.set_timing(args.timing)
        self.overlap = not args.serial
        if self.overlap:
            # the sdust waves stay resident until their queue is empty: leave part of every CU to the other stream
            self.share = args.sdust_share if args.sdust_share > 0 else 76     # (until the probe has run: what it has found on the 3 Gbp workloads since round 5)
            self.share_tuned_for = None
            self.acc2.set_share(self.share)
        # the two large result arrays of this thread (telomere runs, selected windows) travel beside its next kernels
        self.lead_us = float(os.environ.get("CORNETTO_BENCH_LEAD_US", "0"))
        self.handshake = self.overlap and os.environ.get("CORNETTO_BENCH_HANDSHAKE", "1") != "0"
        self.sd_async = self.overlap and os.environ.get("CORNETTO_BENCH_SDUST_ASYNC", "1") == "1"
        self.sd_begun = False
        self.stepped = False
        self.lag_us = float(os.environ.get("CORNETTO_BENCH_SDUST_LAG_US", "0"))
        self.lazy = self.overlap and os.environ.get("CORNETTO_BENCH_LAZY", "1") != "0"
        # cornetto_panel_step (the other thread's three calls as one, two synchronisations instead of five): measured and NOT the default — with its
        # kernels queued back to back the resident sdust waves find no idle issue slots (3.16 Gbp, share 72: 7.55 against 6.98-7.01 ms per step with the
        # three calls; a 1/8 share: 1.15-1.18 against 1.18-1.19, within the noise: the sdust thread bounds the step either way).  CORNETTO_BENCH_FUSED=1.
        self.fused = os.environ.get("CORNETTO_BENCH_FUSED", "0") != "0"
        self.acc.set_lazy(self.lazy)
        self.thr = self.acc.telowin_threshold(0.4, 99.9)
        self.ktime, self.wall = {}, {}
        # the sdust side runs on one persistent worker thread (no thread start inside the timed steps)
        import queue
        import threading
        self.spin_s = float(os.environ.get("CORNETTO_BENCH_SPIN_US", "500")) * 1e-6     # (the worker polls this long for its next job before it blocks)
        self.jobs, self.done = _Handoff(), _Handoff()
        self.worker = threading.Thread(target=self._sdust_worker, daemon=True)
        self.worker.start()

    # ---- workload --------------------------------------------------------------------------------------
    def load(self, profile, scaling=None):
        """generate the rank's inputs in HBM and wrap the contigs this rank owns"""
        args, torch = self.args, self.torch
        self.scaling = scaling or args.scaling
        self.lens = contig_lengths(int(args.gbases * 1e9) if args.gbases > 0 else 0)
        nctg = len(self.lens)
        self.plan = None
        if self.scaling == "strong":
            asm_index = args.assembly_index
            self.job_bases = int(sum(self.lens))
            self.gl_of = lambda own_: np.array(own_, dtype=np.int64)
        else:
            asm_index = args.assembly_index + self.rank
            own = list(range(nctg))
            self.job_bases = int(sum(self.lens)) * self.world
            self.gl_of = lambda own_: np.array(own_, dtype=np.int64) + self.rank * nctg   # global contig ids: assembly-major
        seed = 0xC0FFEE + asm_index
        self.bases, self.offs = make_assembly(torch, self.dev, self.lens, seed, profile)
        self.depth, self.mq = make_coverage(torch, self.dev, self.lens, self.offs, seed)
        torch.cuda.synchronize()
        self.profile = profile
        self.asm = self.asm2 = self.cov = self.cov_halo = None
        if self.scaling == "strong":
            # whole contigs by LPT while they pack within --split-tol of the fair share (the HG002 assembly up to 8 ranks); beyond that the
            # contigs are cut on clean positions and a rank scans pieces with halos (cornetto_amd.dist.SplitPlan).  Every rank asks
            # the same question of the same resident bases: no message
            from cornetto_amd.dist import SplitPlan, make_clean
            seq_of = lambda ci, lo, hi: self.bases[int(self.offs[ci]) + lo:int(self.offs[ci]) + hi].cpu().numpy()      # noqa: E731
            plan = SplitPlan(self.lens, self.world, clean=make_clean(seq_of, b"TTAGGG"), window=2500, inc=50, W=64, min_ctg_len=1000000,
                             tol=args.split_tol)
            if plan.any_split:
                self.plan = plan
                self.wrap_pieces(plan.pieces[self.rank])
                return
            own = [p[0] for p in plan.pieces[self.rank]]                     # (= lpt_partition(lens, world)[rank])
        self.wrap(own)

    def wrap_pieces(self, pieces):
        """strong scaling with contigs above the fair share: this rank's pieces (ctg, start, end, lo, hi) as sequences [lo, hi) of their own"""
        self.unwrap()
        self.stepped = False
        self.pieces = list(pieces)
        self.own = [p[0] for p in pieces]
        self.gl_ctg = np.array(self.own, dtype=np.int64)
        self.lens_own = [p[4] - p[3] for p in pieces]
        self.my_bases = int(sum(p[2] - p[1] for p in pieces))                 # what the rank OWNS (it scans the halos on top: lens_own)
        o = np.array([int(self.offs[p[0]]) + p[3] for p in pieces], dtype=np.int64)
        self.asm = self.acc.asm_wrap(self.bases.data_ptr(), o, np.array(self.lens_own, dtype=np.int64))
        self.asm2 = self.acc2.asm_wrap(self.bases.data_ptr(), o, np.array(self.lens_own, dtype=np.int64))
        self.cov = self.acc.cov_wrap(self.depth.data_ptr(), self.mq.data_ptr(), o, np.array(self.lens_own, dtype=np.int32))
        hr = self.plan.halo_ranges(self.rank)
        if hr:                                                                # the halos alone: their sums leave the totals again
            ho = np.array([int(self.offs[c]) + a for c, a, b in hr], dtype=np.int64)
            self.cov_halo = self.acc.cov_wrap(self.depth.data_ptr(), self.mq.data_ptr(), ho, np.array([b - a for c, a, b in hr], dtype=np.int32))

    def wrap(self, own):
        """(re-)wrap a subset of the resident contigs: what this process scans in a step"""
        self.unwrap()
        self.stepped = False
        self.own = list(own)
        self.gl_ctg = self.gl_of(self.own)
        self.lens_own = [self.lens[i] for i in self.own]
        self.my_bases = int(sum(self.lens_own))
        o = self.offs[self.own] if len(self.own) else np.zeros(0, np.int64)
        self.asm = self.acc.asm_wrap(self.bases.data_ptr(), o, np.array(self.lens_own, dtype=np.int64))
        self.asm2 = self.acc2.asm_wrap(self.bases.data_ptr(), o, np.array(self.lens_own, dtype=np.int64))
        self.cov = self.acc.cov_wrap(self.depth.data_ptr(), self.mq.data_ptr(), o, np.array(self.lens_own, dtype=np.int32))

    def unwrap(self):
        for x in ("asm", "asm2", "cov", "cov_halo"):
            if getattr(self, x, None) is not None:
                getattr(self, x).close()
                setattr(self, x, None)

    def unload(self):
        self.unwrap()
        del self.bases, self.depth, self.mq
        self.torch.cuda.empty_cache()

    # ---- one step --------------------------------------------------------------------------------------
    def _note(self, acc):
        for name, ms in acc.last_timing():
            self.ktime.setdefault(name, []).append(ms)

    def _lap(self, name, t0):
        self.wall.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)

    def _sdust_worker(self):
        while True:
            record = self.jobs.get(self.spin_s)
            if record is None:
                return
            box = {}
            try:
                t0 = time.perf_counter()
                if self.lag_us:                       # (experiment: the other stream's first kernels land on an empty chip first)
                    t_lag = t0 + self.lag_us * 1e-6
                    while time.perf_counter() < t_lag:
                        pass
                if self.sd_begun:                     # (the step's own thread queued the call: cornetto_sdust_asm_begin)
                    box["ivls"] = self.acc2.sdust_end(self.asm2, 20, 64)
                else:
                    box["ivls"] = self.acc2.sdust(self.asm2, 20, 64)
                if record:
                    self._note(self.acc2)
                    self._lap("sdust", t0)
            except BaseException as e:       # re-raised on the main thread
                box["err"] = e
            self.done.put(box)

    def step(self, record, keep=False):
        """The FASTA-side scans (telofind+telowin, sdust) and the coverage stage are independent until the results are
        put together, so a step runs them on two handles (= one HIP stream + workspaces each) — from the second step on from ONE host thread:
        cornetto_sdust_asm_begin() queues the whole sdust call, this thread runs the coverage and telomere stages, cornetto_sdust_asm_end() takes the
        intervals (the first step over an assembly, and CORNETTO_BENCH_SDUST_ASYNC=0: sdust on a second host thread, as in rounds 2-4) —: the
        coverage kernels and every device-to-host copy overlap the long sdust kernel.  (ctypes drops the GIL.)"""
        from cornetto_amd.dist import allreduce_sums, gather_records
        acc, world = self.acc, self.world
        ivls_pre, direct = None, False      # (ivls_pre: an experiment of round 6 ran the first pass over new objects in sequence — slower, DESIGN 8 — and is gone)
        if self.overlap and ivls_pre is None:
            self.acc2.boost(False)                    # this thread's kernels want their share of the chip again
            seq = self.acc2.launch_count() if self.handshake else 0
            t0 = time.perf_counter()
            self.sd_begun = False
            if self.sd_async:
                # the sdust call queued from THIS thread (cornetto_sdust_asm_begin: the whole call in one go where the last step left its counts)
                # and finished by it at the end of the step (cornetto_sdust_asm_end): no hand-over to a thread that has to wake up in front of the
                # kernel launch, none back behind the last result.  Nothing queued (the first step over an assembly): the other thread runs the call
                seq0 = self.acc2.launch_count()
                self.acc2.sdust_begin(self.asm2, 20, 64)
                self.sd_begun = True
                direct = self.acc2.launch_count() != seq0
                if direct and record:
                    self._lap("sdust_begin", t0)
            if not direct:
                self.jobs.put(record)
            if self.handshake and not direct:
                # where the resident sdust waves land decides how much room this thread's kernels find on every CU: let them get there first
                # (8.2 ms per step when they do, 9.2 when they arrive 30 us behind tf_scan: cornetto_accel_launch_count)
                t_end = time.perf_counter() + 1e-3
                while self.acc2.launch_count() == seq and time.perf_counter() < t_end:
                    time.sleep(0)                     # (the sdust thread needs the interpreter lock to get into its call: a spin that keeps it waits 5 ms for it)
                if record:
                    self._lap("handshake", t0)
            if self.lead_us:                          # (experiment: extra lead for the sdust waves)
                t_lead = time.perf_counter() + self.lead_us * 1e-6
                while time.perf_counter() < t_lead:
                    pass
        need_exchange = world > 1 and (self.scaling == "strong" or self.args.allreduce_always)
        if self.fused and self.plan is None:
            # the other thread's whole sequence as ONE call (cornetto_panel_step): totals -> [all-reduce of the three sums] -> thresholds ->
            # selection + telomere scan queued in one go, sized by the last step's counts and checked afterwards: two synchronisations, no
            # Python between the stages (round 5; CORNETTO_BENCH_FUSED=0 brings the three calls back)
            t0 = time.perf_counter()
            xchg = (lambda s_: allreduce_sums(s_, device=self.cdev)) if need_exchange else None
            sums, (self.lo, self.hi), recs_pk, ctg_first, hits, wins = acc.panel_step(self.asm, self.cov, b"TTAGGG", self.thr, 2500, 50, 0.4, 2.5, 0.4, 100000, 1000000,
                                                                                     False, xchg)
            if record:
                self._note(acc)
                self._lap("panel_step", t0)
        else:
            cov_first = os.environ.get("CORNETTO_BENCH_COV_FIRST", "1") != "0"
            if not cov_first:
                t0 = time.perf_counter()
                hits, wins = acc.telo_scan(self.asm, b"TTAGGG", self.thr)
                if record:
                    self._note(acc)
                    self._lap("telo_scan", t0)
            t0 = time.perf_counter()
            sums = acc.cov_prepare(self.cov, 2500, 50)
            if record:
                self._note(acc)
            if self.plan is not None and self.cov_halo is not None:      # pieces: every position counts once
                hs = acc.cov_prepare(self.cov_halo, 2500, 50)
                sums = tuple(int(a) - int(b) for a, b in zip(sums, hs))
            # the one real exchange: the assembly-wide mean depth behind the thresholds (boringbits_main.c:293-294 -> :518-519).
            # Weak scaling keeps every assembly's own mean (N independent assemblies); strong scaling needs the all-reduce.
            if need_exchange:
                sd, sq, n = allreduce_sums(sums, device=self.cdev)
            else:
                sd, sq, n = sums
            mean = int(np.floor(sd / n + 0.5)) if n else 0
            self.lo, self.hi = acc.cov_threshold(0.4, mean), acc.cov_threshold(2.5, mean)
            if record:
                self._lap("cov_prepare", t0)
            t0 = time.perf_counter()
            # the selected windows in packed form (8 B per window + the first record of every contig: include/cornetto_accel.h)
            recs_pk, ctg_first = acc.cov_select_packed(self.cov, self.lo, self.hi, 0.4, 100000, 1000000, False)
            if record:
                self._note(acc)
                self._lap("cov_select", t0)
            if cov_first and os.environ.get("CORNETTO_BENCH_NO_TELO", "0") != "0":
                # (experiment, DESIGN 8: the ceiling of what folding the telomere scan into the sdust waves could gain — a step WITHOUT the scan; not a result)
                hits, wins = np.zeros(0, dtype=cornetto_amd_dt("HIT_DT")), np.zeros(0, dtype=cornetto_amd_dt("WIN_DT"))
            elif cov_first:
                # the coverage stage goes first: its large result copy (8 B per selected window: 60 MB of the 3.16 Gbp assembly) travels beside
                # the telomere kernels instead of being waited for at the end of the step
                t0 = time.perf_counter()
                hits, wins = acc.telo_scan(self.asm, b"TTAGGG", self.thr)
                if record:
                    self._note(acc)
                    self._lap("telo_scan", t0)
        if self.overlap and ivls_pre is None and os.environ.get("CORNETTO_BENCH_BOOST", "1") != "0":
            self.acc2.boost(True)                     # this thread's kernels are through: the waves sdust had left to it join in (cornetto_accel_boost)
        if self.lazy:
            t0 = time.perf_counter()
            acc.wait()                                # the copies of the telomere runs and of the selected windows (cornetto_accel_set_lazy)
            if record:
                self._lap("result_copies", t0)
        recs = recs_pk
        if keep or self.args.gather or self.plan is not None:        # rows with their contig and end, as cornetto_cov_select() returns them
            recs = acc.unpack_regs(recs_pk, ctg_first, self.lens_own, 2500)
        if ivls_pre is not None:
            ivls = ivls_pre
        elif self.overlap and direct:
            t0 = time.perf_counter()
            ivls = self.acc2.sdust_end(self.asm2, 20, 64)
            if record:
                self._note(self.acc2)
                self._lap("sdust_end", t0)            # (what is left to wait for when this thread's own stages are through)
        else:
            if not self.overlap:
                self.jobs.put(record)
            box = self.done.get()
            if "err" in box:
                raise box["err"]
            ivls = box["ivls"]
        if self.plan is not None:
            # pieces: records in contig coordinates, cut down to what each piece owns (part of the step: what a sharded run does before it prints)
            t0 = time.perf_counter()
            pl, rk = self.plan, self.rank
            hits, wins, ivls, recs = pl.own_points(rk, hits, "start"), pl.own_points(rk, wins, "start"), pl.own_intervals(rk, ivls), pl.own_points(rk, recs, "st")
            if record:
                self._lap("own_records", t0)
        gathered = None
        if self.args.gather:                          # optional: all BED/TSV records to rank 0 (RCCL / gloo), global contig order
            t0 = time.perf_counter()
            gathered = [gather_records(arr, self.gl_ctg, device=self.cdev, concat=keep or self.plan is not None) for arr in (hits, wins, ivls, recs)]
            if self.plan is not None and gathered[0] is not None:
                from cornetto_amd.dist import order_records, stitch_intervals
                gathered = [order_records(gathered[0], ("strand", "start")), order_records(gathered[1], ("start",)), stitch_intervals(gathered[2]),
                            order_records(gathered[3], ("st",))]
            if record:
                self._lap("gather", t0)
        self.counts = [len(hits), len(wins), len(ivls), len(recs)]
        self.stepped = True
        if keep:
            return (hits, wins, ivls, recs), gathered
        return None

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def tune_share(self):
        """untimed, part of the warm-up: which share of a CU's wave slots the resident sdust waves take beside the other stream depends on
        what the sequence makes of sdust (uniform: the two host threads are balanced at 70 %; repeat-rich: sdust is three times the rest and
        wants nearly all of the chip).  Three steps at each of a few shares, the best is kept (cornetto_accel_set_share); every rank
        tunes for itself."""
        if not self.overlap or self.args.sdust_share > 0:
            return
        if self.share_tuned_for == (self.profile, self.scaling, self.my_bases):
            return                                       # (probed once per resident workload)
        self.share_tuned_for = (self.profile, self.scaling, self.my_bases)
        for _ in range(2):                             # (the result pools of a new workload grow in its first steps: not the share's doing)
            self.step(False)
        best, seen = None, {}
        for sh in (85, 80, 76, 72, 68, 92, 100, 85):   # (the first one twice: the first candidate measured is the one that pays for what is still warming up)
            self.acc2.set_share(sh)
            self.step(False)
            self.torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                self.step(False)
            self.torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 4 / 6      # (six steps per candidate since round 5: four left the choice to the noise; reported per four as before)
            seen[sh] = min(dt, seen.get(sh, dt))
        for sh, dt in seen.items():
            if best is None or dt < best[0]:
                best = (dt, sh)
        self.share = best[1]
        self.share_probe_ms = {str(k): round(v / 4 * 1e3, 3) for k, v in seen.items()}
        self.acc2.set_share(self.share)

    def timed(self, steps, warmup):
        """`warmup` untimed steps, then exactly `steps` steps between two fences; max over ranks -> seconds"""
        for _ in range(warmup):
            self.step(False)
        self.tune_share()
        self.ktime.clear()
        self.wall.clear()
        self.fence()
        t0 = time.perf_counter()
        counts, marks = [], [t0]
        for _ in range(steps):
            self.step(True)
            counts.append(tuple(self.counts))
            marks.append(time.perf_counter())         # (a step has delivered its results to the host when it returns)
        self.fence()
        elapsed = time.perf_counter() - t0
        self.elapsed_local = elapsed
        self.step_ms = [(b - a) * 1e3 for a, b in zip(marks, marks[1:])]
        if self.world > 1:
            el = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.cdev)
            self.dist.all_reduce(el, op=self.dist.ReduceOp.MAX)
            elapsed = float(el.item())
        if len(set(counts)) != 1:
            raise SystemExit("result counts changed between steps: %r" % sorted(set(counts)))
        return elapsed

    def serial_kernel_times(self):
        """one extra, untimed pass with the stages serial on one stream: uncontended per-kernel durations (the HBM-bound
        kernels of the high-priority stream are slowed by the co-running sdust kernel in the timed steps)"""
        if not self.overlap:
            return None
        timed = {k: list(v) for k, v in self.ktime.items()}
        wall = {k: list(v) for k, v in self.wall.items()}
        self.ktime.clear()
        self.overlap = False
        self.acc2.set_share(100)
        self.acc.set_timing(2)
        self.acc2.set_timing(2)
        self.step(True)
        self.acc.set_timing(self.args.timing)
        self.acc2.set_timing(self.args.timing)
        self.acc2.set_share(self.share)
        self.overlap = True
        serial, self.ktime, self.wall = self.ktime, timed, wall
        return serial

    def close(self):
        self.jobs.put(None)
        self.worker.join()
        self.acc.close()
        self.acc2.close()
        if self.world > 1:
            self.dist.destroy_process_group()


def kernel_table(R, serial, n_bases):
    kavg = {k: float(np.mean(v)) for k, v in R.ktime.items()}
    # algorithmic bytes per launch: sdust_kernel and tf_scan read 1 B/base,
    # cov_blocks reads 4 B/base (u16 depth + u16 mq)
    alg = {"sdust_kernel": 1.0 * n_bases, "tf_scan": 1.0 * n_bases, "cov_blocks": 4.0 * n_bases}
    kern = {}
    for k in sorted(set(kavg) | set(serial or {})):
        ms = kavg.get(k)
        kern[k] = {"ms": round(ms, 4)} if ms is not None else {}     # (the timed steps carry events around the main kernels only: --timing)
        if ms and k in alg:
            kern[k]["algorithmic_GBps"] = round(alg[k] / (ms * 1e-3) / 1e9, 2)
        if serial and k in serial:
            sm = float(np.mean(serial[k]))
            kern[k]["ms_uncontended"] = round(sm, 4)
            if k in alg and sm > 0:
                kern[k]["algorithmic_GBps_uncontended"] = round(alg[k] / (sm * 1e-3) / 1e9, 2)
    return kavg, alg, kern


def _hash_lines(chunks):
    """(blake2b-128 of the concatenated bytes, number of newlines)"""
    import hashlib
    h, n = hashlib.blake2b(digest_size=16), 0
    buf = []
    for c in chunks:
        buf.append(c)
        if len(buf) >= 65536:
            b = b"".join(buf)
            h.update(b)
            n += b.count(b"\n")
            buf = []
    b = b"".join(buf)
    h.update(b)
    n += b.count(b"\n")
    return h.hexdigest(), n


def e2e_cli(R, cornetto_amd):
    """the C CLI end to end on the rank's assembly written as a single-line FASTA into memory-backed /dev/shm (or /tmp):
    process start, HIP initialisation, file read, record framing on the device, scan, printing — the file- and PCIe-bound
    number that belongs beside the HBM-resident one"""
    import subprocess
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    path = os.path.join(shm, "cornetto_bench_asm.%d.fa" % os.getpid())
    out = {"fasta_bytes": 0, "where": shm}
    try:
        fs = os.statvfs(shm)
        if fs.f_bavail * fs.f_frsize < 2 * R.my_bases + (1 << 30):
            return {"skipped": "not enough free space under %s for a %d-byte FASTA" % (shm, R.my_bases)}
        hb = R.bases.cpu().numpy()
        with open(path, "wb") as f:
            for i in R.own:
                f.write(b">ptg%06dl\n" % i)
                f.write(memoryview(hb[int(R.offs[i]):int(R.offs[i]) + int(R.lens[i])]))
                f.write(b"\n")
        del hb
        out["fasta_bytes"] = os.path.getsize(path)
        # what the CLI must print: the records of a step over the same resident assembly (the step's records are compared with the reference's own
        # functions in this run: "parity"), put into text with the reference's printf formats (sdust.c:201 "%s\t%d\t%d\n"; find_telomere.c:51,56
        # "%s\t%zu\t0\t%zu" + "\t%zu\t%zu\n", strand column 1 for the reverse search :66,71)
        (hits, _wins, ivls, _recs), _ = R.step(False, keep=True)
        names = [b"ptg%06dl" % i for i in R.own]
        lens_own = R.lens_own
        want = {"sdust": _hash_lines(b"%s\t%d\t%d\n" % (names[c], a, b) for c, a, b in zip(ivls["ctg"].tolist(), ivls["start"].tolist(), ivls["finish"].tolist())),
                "telofind": _hash_lines(b"%s\t%d\t%d\t%d\t%d\t%d\n" % (names[c], lens_own[c], sd, a, b, b - a)
                                        for c, sd, a, b in zip(hits["ctg"].tolist(), hits["strand"].tolist(), hits["start"].tolist(), hits["end"].tolist()))}
        del hits, ivls, _wins, _recs
        for sub in ("sdust", "telofind"):
            best, nbytes, got = None, 0, None
            for _ in range(3):                         # (process start, HIP initialisation and exit vary by tens of milliseconds from run to run: the best of three)
                t0 = time.perf_counter()
                p = subprocess.run([cornetto_amd.CLI_PATH, sub, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   env=dict(os.environ, CORNETTO_DEVICE=str(R.local_dev)))
                dt = time.perf_counter() - t0
                if p.returncode != 0:
                    out[sub] = {"error": p.stderr[-300:].decode("replace")}
                    best = None
                    break
                nbytes = len(p.stdout)
                got = _hash_lines([p.stdout])
                best = dt if best is None else min(best, dt)
            if best is not None:
                out[sub] = {"wall_s": round(best, 3), "gbases_s": round(R.my_bases / best / 1e9, 3), "stdout_bytes": nbytes,
                            "stdout_equals_step_records": bool(got == want[sub]), "stdout_lines": want[sub][1]}
        # and the CLI beside the unmodified reference binary (oracle/_ref/cornetto, where it was built) on a FASTA both finish in seconds: the
        # smallest contigs of the assembly up to 40 Mbases plus the smallest contig of at least 12 Mb (telomere arrays at both ends, every planted
        # feature kind), byte for byte
        if os.path.exists(REF_BIN):
            sub_path = path + ".sub"
            order = sorted(range(len(R.own)), key=lambda li: R.lens_own[li])
            take, tot = [], 0
            for li in order:
                if tot + R.lens_own[li] > 40_000_000:
                    break
                take.append(li)
                tot += R.lens_own[li]
            mid = [li for li in order if R.lens_own[li] >= 12_000_000 and li not in take][:1]
            take = sorted(take + mid)
            hb = R.bases.cpu().numpy()
            with open(sub_path, "wb") as f:
                for li in take:
                    i = R.own[li]
                    f.write(b">ptg%06dl\n" % i)
                    f.write(memoryview(hb[int(R.offs[i]):int(R.offs[i]) + int(R.lens[i])]))
                    f.write(b"\n")
            del hb
            try:
                same = {}
                for sub in ("sdust", "telofind"):
                    pr = subprocess.run([REF_BIN, sub, sub_path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                    pg = subprocess.run([cornetto_amd.CLI_PATH, sub, sub_path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                        env=dict(os.environ, CORNETTO_DEVICE=str(R.local_dev)))
                    same[sub] = bool(pr.returncode == 0 and pg.returncode == 0 and pr.stdout == pg.stdout and len(pr.stdout) > 0)
                    same[sub + "_bytes"] = len(pr.stdout)
                out["same_as_reference"] = dict(same, ok=bool(same["sdust"] and same["telofind"]), contigs=len(take), bases=int(sum(R.lens_own[li] for li in take)),
                                                what="stdout of `cornetto sdust` / `telofind` of this build and of oracle/_ref/cornetto (the unmodified reference) on the same FASTA, byte for byte")
            finally:
                os.remove(sub_path)
        # where the time of `cornetto sdust` goes (CORNETTO_CLI_TRACE: milliseconds since process start at each point; the pinned piece
        # can only be allocated once the HIP runtime is up, so "pinned piece allocated" = runtime initialisation + code object + pinning)
        t0 = time.perf_counter()
        p = subprocess.run([cornetto_amd.CLI_PATH, "sdust", path], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                           env=dict(os.environ, CORNETTO_DEVICE=str(R.local_dev), CORNETTO_CLI_TRACE="1"))
        tr = []
        for l in p.stderr.decode(errors="replace").splitlines():
            if l.startswith("[cli trace]"):
                f = l[len("[cli trace]"):].rsplit(None, 2)
                tr.append([f[0].strip(), float(f[1])])
        if tr:
            out["sdust_trace_ms"] = tr[:4] + ([["..."]] if len(tr) > 8 else []) + tr[-4:] if len(tr) > 8 else tr
            out["sdust_trace_wall_s"] = round(time.perf_counter() - t0, 3)
            # the floor under the CLI's wall time: the HIP runtime and the handle must be up before the first pinned slab exists
            for name, key in (("device open", "device_open_s"), ("text on the device", "text_on_device_s"), ("records framed", "records_framed_s"),
                              ("scanned and printed", "scanned_and_printed_s")):
                hit = [t for n_, t in tr if n_ == name]
                if hit and "sdust" in out and isinstance(out["sdust"], dict):
                    out["sdust"][key] = round(hit[0] / 1e3, 3)
    except Exception as e:                               # the e2e figure is an extra: never fail the bench line over it
        out["error"] = repr(e)
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    return out


def profile_parity(R, budget_bases=45_000_000):
    """parity of the step's records on the workload currently loaded against the reference's own functions (oracle/_ref; the oracle port where
    that is absent) on the two smallest contigs that hold planted satellite arrays (make_assembly plants them in the contigs of at least 12 Mb),
    all four stages, record for record: the repeat-rich profiles print numbers of their own in the line, so they carry a check of their own"""
    big = sorted((R.lens_own[li], li) for li in range(len(R.lens_own)) if R.lens_own[li] >= 12_000_000)
    if not big:
        big = sorted(((R.lens_own[li], li) for li in range(len(R.lens_own))), reverse=True)[:2]
    pick, tot = [], 0
    for n, li in big:
        if pick and (len(pick) >= 2 or tot + n > budget_bases):
            break
        pick.append(li)
        tot += n
    pick.sort()
    t0 = time.perf_counter()
    base, results = cpu_reference_leg(R.bases, R.depth, R.mq, R.offs, R.lens, R.own, 0, pick=pick)
    last, _ = R.step(False, keep=True)
    par = check_parity(results, last, R.lo, R.hi, 0.4, 1000000, R.lens_own)
    par["against"] = "the reference's own functions (oracle/_ref)" if base["kind"] == "reference" else "the oracle port"
    par["contigs_local"] = pick
    par["cpu_s"] = round(time.perf_counter() - t0, 2)
    return par


def profile_leg(R, steps, parity=True):
    """ms/step and the sdust kernel on the workload currently loaded, plus the kernel's own statistics run"""
    el = R.timed(steps, 1)
    kavg = {k: float(np.mean(v)) for k, v in R.ktime.items()}
    out = {"ms_per_step": round(el / steps * 1e3, 3), "gbases_s": round(R.job_bases / (el / steps) / 1e9, 3), "sdust_share_percent": getattr(R, "share", None),
           "sdust_share_probe_ms": getattr(R, "share_probe_ms", None),
           "sdust_kernel_ms": round(kavg.get("sdust_kernel", 0.0), 3), "tf_scan_ms": round(kavg.get("tf_scan", 0.0), 3),
           "cov_blocks_ms": round(kavg.get("cov_blocks", 0.0), 3),
           "results": dict(zip(("telomere_runs", "telomere_windows", "sdust_intervals", "selected_cov_windows"), R.counts))}
    serial = R.serial_kernel_times()
    if serial and "sdust_kernel" in serial:
        out["sdust_kernel_ms_uncontended"] = round(float(np.mean(serial["sdust_kernel"])), 3)
    st = R.acc2.sdust_stats(R.asm2, 20, 64) if hasattr(R.acc2, "sdust_stats") else None
    if st:
        out["sdust_stats"] = st
    if parity:
        out["parity"] = profile_parity(R)
    return out


def _shm_dir():
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"


def e2e_noboringbits(R, torch, cornetto_amd, mlines=100.0):
    """BASELINE config 3 end to end: the C CLI `cornetto noboringbits cov-total.bg -q cov-mq20.bg` on two per-base bedgraph FILES
    (memory-backed): process start, HIP initialisation, file read, H2D, text parse on the device (the reference's fscanf loop is 95 %
    of its wall time, boringbits_main.c:204-287), window stage, selection, printing"""
    import subprocess
    shm = _shm_dir()
    n = int(mlines * 1e6)
    out = {"where": shm}
    paths = [os.path.join(shm, "cornetto_bench_cov_%s.%d.bg" % (t, os.getpid())) for t in ("total", "mq20")]
    try:
        fs = os.statvfs(shm)
        while n >= 10_000_000 and fs.f_bavail * fs.f_frsize < 2 * 34 * n + (2 << 30):
            n //= 2
        if n < 10_000_000:
            return {"skipped": "not enough free space under %s" % shm}
        for path, seed, mq in zip(paths, (11, 12), (False, True)):
            t = make_bedgraph_text(torch, R.dev, n, 11, mq)       # (same seed: mq differs from depth only in its segments)
            t.cpu().numpy().tofile(path)
            del t
        torch.cuda.empty_cache()
        nbytes = sum(os.path.getsize(p) for p in paths)
        best, so = None, b""
        for _ in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([cornetto_amd.CLI_PATH, "noboringbits", paths[0], "-q", paths[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               env=dict(os.environ, CORNETTO_DEVICE=str(R.local_dev)))
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": p.stderr[-300:].decode("replace")}
            so = p.stdout
            best = dt if best is None else min(best, dt)
        # the same with the text cut into two shares of whole contigs (CORNETTO_DEVICES: one host thread, handle and reader pool per listed
        # device — here the SAME GPU twice: what the cut and the second handle cost; on a node every share has its own PCIe link)
        t0 = time.perf_counter()
        p2 = subprocess.run([cornetto_amd.CLI_PATH, "noboringbits", paths[0], "-q", paths[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, CORNETTO_DEVICES="%d,%d" % (R.local_dev, R.local_dev)))
        out["sharded_ingest_same_gpu_twice"] = {"wall_s": round(time.perf_counter() - t0, 3), "same_stdout": p2.returncode == 0 and p2.stdout == so,
                                                "sharded": b"sharded ingest" in p2.stderr}
        # the same CLI beside the unmodified reference binary on the leading 10 M positions (= the first contig) of both files, stdout byte for
        # byte and the exit status (the reference's fscanf loop reads 20 M lines in a few seconds; boringbits_main.c:204-287)
        if os.path.exists(REF_BIN):
            sl = [p_ + ".slice" for p_ in paths]
            try:
                nsl = min(n, 10_000_000)
                for src_, dst_ in zip(paths, sl):
                    with open(src_, "rb") as f, open(dst_, "wb") as g_:
                        g_.write(f.read(34 * nsl))
                t0 = time.perf_counter()
                pr = subprocess.run([REF_BIN, "noboringbits", sl[0], "-q", sl[1]], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                t_ref = time.perf_counter() - t0
                t0 = time.perf_counter()
                pg = subprocess.run([cornetto_amd.CLI_PATH, "noboringbits", sl[0], "-q", sl[1]], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                    env=dict(os.environ, CORNETTO_DEVICE=str(R.local_dev)))
                t_gpu = time.perf_counter() - t0
                out["same_as_reference"] = {"ok": bool(pr.returncode == 0 and pg.returncode == 0 and pr.stdout == pg.stdout and len(pr.stdout) > 0),
                                            "positions": nsl, "stdout_bytes": len(pr.stdout), "reference_wall_s": round(t_ref, 2), "wall_s": round(t_gpu, 2),
                                            "what": "stdout of `cornetto noboringbits` of this build and of oracle/_ref/cornetto on the same two bedgraph files, byte for byte"}
            finally:
                for p_ in sl:
                    try:
                        os.remove(p_)
                    except OSError:
                        pass
        out.update({"positions": n, "contigs": (n + 9_999_999) // 10_000_000, "text_bytes": nbytes, "wall_s": round(best, 3), "text_GBps": round(nbytes / best / 1e9, 2),
                    "gbases_s": round(n / best / 1e9, 3), "stdout_bytes": len(so), "stdout_lines": so.count(b"\n"),
                    "command": "cornetto noboringbits cov-total.bg -q cov-mq20.bg (defaults: -w 2500 -i 50 -L 0.4 -H 2.5 -Q 0.4)"})
    except Exception as e:                               # an extra: never fail the bench line over it
        out["error"] = repr(e)
    finally:
        for path in paths:
            try:
                os.remove(path)
            except OSError:
                pass
    return out


def reads_shares(piece_lens, n_pieces, world):
    """Config 5's read-level shard: the stream is `n_pieces` pieces (piece j = the generated piece j % len(piece_lens)), cut into
    `world` contiguous shares at READ boundaries by cumulative bases (share r holds the reads whose first base has a cumulative
    index in [r B / world, (r + 1) B / world)).  -> per rank a list of (piece slot, first read, one past the last read), no empty
    segments; every read of the stream is in exactly one segment and the segments of the ranks, in rank order, are the stream."""
    k = len(piece_lens)
    tot = [int(np.sum(l)) for l in piece_lens]
    B = sum(tot[j % k] for j in range(n_pieces))
    shares = [[] for _ in range(world)]
    base = 0
    for j in range(n_pieces):
        lens = np.asarray(piece_lens[j % k], dtype=np.int64)
        start = base + np.concatenate([[0], np.cumsum(lens)[:-1]])          # cumulative index of every read's first base
        owner = np.minimum(world - 1, (start * world) // B)                  # non-decreasing
        for r in range(int(owner[0]), int(owner[-1]) + 1):
            lo, hi = int(np.searchsorted(owner, r, "left")), int(np.searchsorted(owner, r, "right"))
            if hi > lo:
                shares[r].append((j % k, lo, hi))
        base += tot[j % k]
    return shares


def reads_leg(R, torch, dist, cornetto_amd, gbases):
    """BASELINE config 5: FASTQ text in pinned host memory -> records framed on the device with the length test of
    `seq -m 10000` (src/seq.c:120) -> reads packed in HBM -> sdust per read (src/sdust/sdust.c:196-203) -> intervals on the host.
    Two host threads with a handle (= HIP stream) each take the work items alternately, so the H2D copy of one item runs beside
    the kernels of the other.  One rank: the items are the pieces of the stream.  N ranks: the SAME stream cut into N shares at
    read boundaries by cumulative bases (reads_shares; no data-path collective: every rank owns the output of its reads), each
    rank streams its share over its own PCIe link, wall = barrier to barrier, max over the ranks; the ranks' counts must add up
    to what rank 0 gets for the whole pieces.  Rank 0 also runs the first ~200 Mbases through the reference's own
    `seq -m 10000 | sdust` (oracle/_ref) on one core and compares line for line.  Every rank calls this; rank 0 gets the dict."""
    import ctypes as C
    import subprocess
    import threading
    world, rank = R.world, R.rank
    out = {"min_len": 10000, "sdust": "-w 64 -t 20"}
    L = cornetto_amd.lib()
    pins = []

    def everyone(flag):
        """True when `flag` holds on every rank (a rank that cannot go on must not leave the others in a barrier)"""
        if world == 1:
            return bool(flag)
        fl = [None] * world
        dist.all_gather_object(fl, bool(flag))
        return all(fl)

    try:
        piece_bases = min(1.0e9, max(1.0e6, gbases * 1e9 / 2))     # (small --reads-gbases: tests)
        pieces = []
        for pi in range(2):
            text, lens, off = make_fastq_piece(torch, R.dev, piece_bases, 500 + pi)
            n = int(text.numel())
            pin = L.cornetto_pinned_alloc(n + 64)
            if pin:
                pins.append(pin)
                host = text.cpu().numpy()
                C.memmove(pin, host.ctypes.data, n)
                pieces.append({"pin": pin, "n": n, "lens": lens, "off": np.concatenate([off, [n]]).astype(np.int64), "host": host if (pi == 0 and rank == 0) else None})
            del text
            if not everyone(bool(pin)):
                return {"skipped": "no pinned host memory for a %d-byte piece" % n}
        torch.cuda.empty_cache()
        n_pieces = max(2, int(np.ceil(gbases * 1e9 / piece_bases)))
        if world > 1:
            n_pieces = max(n_pieces, 2 * world)        # at least two pieces' worth of text per rank
        accs = [cornetto_amd.Accel(R.local_dev), cornetto_amd.Accel(R.local_dev)]
        for a in accs:
            a.set_timing(1)

        def item(slot, lo, hi):
            p = pieces[slot]
            return (p["pin"] + int(p["off"][lo]), int(p["off"][hi] - p["off"][lo]), hi - lo, int(p["lens"][lo:hi].sum()))

        whole = [item(s, 0, len(pieces[s]["lens"])) for s in range(2)]
        if world == 1:
            items = [whole[j % 2] for j in range(n_pieces)]
        else:
            segs = reads_shares([p["lens"] for p in pieces], n_pieces, world)[rank]
            # (a share of one or two long segments is cut once more so that both host threads have items of similar size)
            items = []
            for s, lo, hi in segs:
                if hi - lo >= 2 and len(segs) < 4:
                    mid = lo + int(np.searchsorted(np.cumsum(pieces[s]["lens"][lo:hi]), int(pieces[s]["lens"][lo:hi].sum()) // 2)) + 1
                    mid = min(max(mid, lo + 1), hi - 1)
                    items += [item(s, lo, mid), item(s, mid, hi)]
                else:
                    items.append(item(s, lo, hi))
        ktime = [0.0, 0.0]
        err = []

        def work(w, todo, res):
            try:
                for j in range(w, len(todo), 2):
                    addr, nb, nreads, _ = todo[j]
                    recs, used, plain, reads = accs[w].fastq_split((addr, nb), final=True, min_len=10000, want_reads=True)
                    iv = accs[w].sdust(reads, 20, 64)
                    ktime[w] += sum(ms for k, ms in accs[w].last_timing() if k == "sdust_kernel")
                    reads.close()
                    if not plain or used != nb or len(recs) != nreads:
                        raise RuntimeError("item %d: framing fell back (plain %r, used %d of %d, %d of %d records)" % (j, plain, used, nb, len(recs), nreads))
                    res[j] = (int(recs["keep"].sum()), int(recs["len"][recs["keep"] == 1].sum()), len(iv), digest([iv]))
            except BaseException as e:
                err.append(repr(e))

        def run(todo):
            res = [None] * len(todo)
            ktime[0] = ktime[1] = 0.0
            th = [threading.Thread(target=work, args=(w, todo, res)) for w in range(2)]
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            if world > 1:
                dist.barrier()
            return time.perf_counter() - t0, res

        # the first pass warms the workspaces up; on rank 0 of a sharded run it is also what the whole pieces give (the check below)
        _, res_whole = run(whole)
        if not everyone(not err):
            return {"error": err[0] if err else "another rank failed"}
        wall, res = run(items)
        if not everyone(not err):
            return {"error": err[0] if err else "another rank failed"}
        mine = {"items": len(items), "text_bytes": sum(i[1] for i in items), "reads_in": sum(i[2] for i in items), "bases_in": sum(i[3] for i in items),
                "reads_kept": sum(r[0] for r in res), "bases_kept": sum(r[1] for r in res), "sdust_intervals": sum(r[2] for r in res),
                "wall_s": wall, "sdust_kernel_ms_sum": ktime[0] + ktime[1]}
        allr = [mine]
        if world > 1:
            allr = [None] * world
            dist.all_gather_object(allr, mine)
        for a in accs:
            a.close()
        if rank != 0:
            return None
        wall = max(x["wall_s"] for x in allr)
        tot = {k: sum(x[k] for x in allr) for k in ("text_bytes", "reads_in", "bases_in", "reads_kept", "bases_kept", "sdust_intervals")}
        exp = {"reads_in": sum(whole[j % 2][2] for j in range(n_pieces)), "bases_in": sum(whole[j % 2][3] for j in range(n_pieces)),
               "reads_kept": sum(res_whole[j % 2][0] for j in range(n_pieces)), "bases_kept": sum(res_whole[j % 2][1] for j in range(n_pieces)),
               "sdust_intervals": sum(res_whole[j % 2][2] for j in range(n_pieces))}
        if world == 1:
            same = all(res[j][3] == res_whole[j % 2][3] for j in range(n_pieces))
        else:
            same = all(tot[k] == exp[k] for k in exp)
        out.update({"pieces": n_pieces, "ranks": world})
        out.update(tot)
        out.update({"wall_s": round(wall, 3), "gbases_s_text_in": round(tot["bases_in"] / wall / 1e9, 3), "gbases_s_kept": round(tot["bases_kept"] / wall / 1e9, 3),
                    "pcie_GBps": round(tot["text_bytes"] / wall / 1e9, 2), "sdust_kernel_ms_sum": round(sum(x["sdust_kernel_ms_sum"] for x in allr), 2)})
        if world == 1:
            out["repeats_identical"] = bool(same)
            out["how"] = ("2 host threads x 1 handle each, pieces of ~%.2g Gbase from pinned memory" % (piece_bases / 1e9) + " (2 B of text per base): cornetto_fastq_split(min_len = 10000) + cornetto_sdust_asm; "
                          "the two distinct pieces are streamed alternately")
        else:
            out["shares_add_up"] = bool(same)
            out["per_rank"] = [{"bases_in": x["bases_in"], "reads_in": x["reads_in"], "items": x["items"], "wall_s": round(x["wall_s"], 3)} for x in allr]
            out["how"] = ("one stream of %d pieces of ~1 Gbase (less with a small --reads-gbases) cut into %d shares at read boundaries by cumulative bases (src/seq.c:116-129 per read; reads_shares); every rank: "
                          "2 host threads x 1 handle, its share from its own pinned memory over its own PCIe link: cornetto_fastq_split(min_len = 10000) + cornetto_sdust_asm; "
                          "no collective in the data path; wall = barrier to barrier, the slowest rank; the ranks' read / kept / interval counts are compared with "
                          "those of the whole pieces on rank 0" % (n_pieces, world))
        if not same:
            out["parity"] = {"ok": False, "what": "the shares do not add up to the whole pieces" if world > 1 else "a repeated piece gave other intervals", "got": tot, "expected": exp}
            return out
        # ---- the reference on a sample: seq -m 10000 | sdust, one core ------------------------------------------
        ref_cli = os.path.join(ROOT, "oracle", "_ref", "cornetto")
        if os.path.exists(ref_cli):
            p0 = pieces[0]
            j = int(np.searchsorted(np.cumsum(p0["lens"]), 200e6)) + 1
            cut = int(p0["off"][j]) if j < len(p0["lens"]) else p0["n"]
            shm = _shm_dir()
            f1 = os.path.join(shm, "cornetto_bench_reads.%d.fq" % os.getpid())
            f2 = os.path.join(shm, "cornetto_bench_reads_m.%d.fq" % os.getpid())
            try:
                p0["host"][:cut].tofile(f1)
                t0 = time.perf_counter()
                with open(f2, "wb") as fh:
                    subprocess.run([ref_cli, "seq", "-m", "10000", f1], stdout=fh, stderr=subprocess.DEVNULL, check=True)
                t1 = time.perf_counter()
                exp = subprocess.run([ref_cli, "sdust", f2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
                t2 = time.perf_counter()
                acc = cornetto_amd.Accel(R.local_dev)
                sample = np.ascontiguousarray(p0["host"][:cut])
                recs, used, plain, reads = acc.fastq_split(sample, final=True, min_len=10000, want_reads=True)
                iv = acc.sdust(reads, 20, 64)
                reads.close()
                acc.close()
                kept = recs[recs["keep"] == 1]
                names = [bytes(sample[int(h) + 1:int(h) + 1 + int(nl)]) for h, nl in zip(kept["head"], kept["name_len"])]
                got = b"".join(b"%s\t%d\t%d\n" % (names[int(x["ctg"])], x["start"], x["finish"]) for x in iv)
                sb = int(p0["lens"][:j].sum())
                out["parity"] = {"ok": bool(got == exp), "sample_bases": sb, "reads_kept": len(kept), "intervals": len(iv),
                                 "against": "oracle/_ref/cornetto seq -m 10000 | oracle/_ref/cornetto sdust (the unmodified reference)"}
                out["cpu_reference"] = {"cores": 1, "seq_s": round(t1 - t0, 2), "sdust_s": round(t2 - t1, 2), "gbases_s": round(sb / (t2 - t0) / 1e9, 4), "sample_bases": sb}
            finally:
                for f in (f1, f2):
                    try:
                        os.remove(f)
                    except OSError:
                        pass
    except Exception as e:                               # an extra: never fail the bench line over it (a parity mismatch does)
        if world > 1:
            raise                                        # (a rank that drops out of the collectives would hang the others)
        out["error"] = repr(e)
    finally:
        for pin in pins:
            L.cornetto_pinned_free(pin)
    return out


def emulate_ranks(R, ns, steps, t1_ms):
    """Strong scaling modelled on ONE GPU: for N in ns the contigs are split exactly as an N-rank run splits them
    (cornetto_amd.dist.lpt_partition), and every rank's share is run as its own resident workload, one after the other, with
    the normal two-stream step.  A real N-GPU step ends when its slowest rank ends (plus one 24-byte all-reduce, not modelled):
    step_ms = max over the shares; efficiency = t1 / (N * step_ms).  What this measures is how well the fixed costs of a step
    (small launches, event pairs, host hand-offs, result copies) scale down with the share — the part of the curve one GPU can show."""
    from cornetto_amd.dist import lpt_partition
    out = {"modelled": True, "t1_ms": round(t1_ms, 3), "steps_per_share": steps,
           "what": "per-rank shares of the LPT contig partition run one after the other on one GPU (the better of two batches of steps_per_share steps each); step = slowest share; the 3 x int64 all-reduce is not modelled"}
    full = list(range(len(R.lens)))
    for n in ns:
        parts = lpt_partition(R.lens, n)
        per, bases, stages, nctg = [], [], [], []
        order = list(range(len(parts)))
        if os.environ.get("CORNETTO_BENCH_EMU_REVERSE"):      # (diagnosis: is a slow share slow, or just measured first?)
            order.reverse()
        for ri in order:
            own = parts[ri]
            R.wrap(own)
            # (two warm-up steps: the second sdust call of a chunk table is the first with its long chunks ordered first.  The better of two
            # batches: whichever shares are measured first after the full-size run come out 0.15-0.2 ms slower than the same shares measured
            # later — the order of measurement, not the share; a rank of a real run is in its steady state)
            el = min(R.timed(steps, 2), R.timed(steps, 0))
            per.append(round(el / steps * 1e3, 3))
            bases.append(R.my_bases)
            nctg.append(len(own))
            stages.append({k: round(float(np.mean(v)), 3) for k, v in R.wall.items()})
        # (the share measured first once more at the end: it is the one that pays for the change from 8 ms steps to 1 ms steps)
        R.wrap(parts[order[0]])
        again = round(min(R.timed(steps, 2), R.timed(steps, 0)) / steps * 1e3, 3)
        if again < per[0]:
            per[0] = again
            stages[0] = {k: round(float(np.mean(v)), 3) for k, v in R.wall.items()}
        mx = max(per)
        out[str(n)] = {"per_rank_ms": per, "bases_per_rank": bases, "contigs_per_rank": nctg, "step_ms": mx, "efficiency": round(t1_ms / (n * mx), 4),
                       "gbases_s": round(sum(bases) / (mx * 1e-3) / 1e9, 2), "stage_wall_ms_slowest": stages[per.index(mx)], "stage_wall_ms_fastest": stages[per.index(min(per))]}
    R.wrap(full)
    return out


def device_identity(torch, idx):
    """what tells two GPUs of a node apart: PCI domain:bus:device, else the uuid, else the ordinal"""
    p = torch.cuda.get_device_properties(idx)
    for names in (("pci_domain_id", "pci_bus_id", "pci_device_id"),):
        if all(hasattr(p, n) for n in names):
            return "%04x:%02x:%02x" % tuple(int(getattr(p, n)) for n in names)
    if hasattr(p, "uuid"):
        return str(p.uuid)
    return "ordinal:%d" % idx


def collectives_info(R, dist, torch, elapsed_local, steps):
    """who ran where and what the exchange costs: backend, world size, (host, device) of every rank, per-rank step times,
    the latency of the 3 x int64 all-reduce.  Rank 0 gets the dict; every rank gets `shared` (two ranks on one device)."""
    import socket
    me = (socket.gethostname(), device_identity(torch, R.local_dev), round(elapsed_local / steps * 1e3, 3))
    allr = [None] * R.world
    dist.all_gather_object(allr, me)
    t = torch.tensor([1, 2, 3], dtype=torch.int64, device=R.cdev)
    for _ in range(5):
        dist.all_reduce(t)
    if R.cdev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        dist.all_reduce(t)
    if R.cdev.type == "cuda":
        torch.cuda.synchronize()
    ar_us = (time.perf_counter() - t0) / reps * 1e6
    devs = [(h, d) for h, d, _ in allr]
    shared = len(set(devs)) != len(devs)
    info = {"backend": dist.get_backend(), "world": dist.get_world_size(), "devices": ["%s/%s" % d for d in devs],
            "distinct_devices": not shared, "per_rank_ms": [x[2] for x in allr], "allreduce_3xi64_us": round(ar_us, 1),
            "data_path": "strong: one all-reduce of 3 x int64 per step (boringbits_main.c:293-294 -> :518-519); weak: none; --gather: all_gather of counts + gather of the record arrays to rank 0"}
    return info, shared


def measure(R, args, scaling, profile, steps, warmup, gather):
    """load (when the scaling or profile differs from what is resident), time, return (elapsed seconds, serial kernel times)"""
    if getattr(R, "bases", None) is None or R.scaling != scaling or R.profile != profile:
        if getattr(R, "bases", None) is not None:
            R.unload()
        R.load(profile, scaling)
    keep = args.gather
    args.gather = gather
    try:
        el = R.timed(steps, warmup)
    finally:
        args.gather = keep
    return el


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (`python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>`), pass rank 0's JSON
    line through, return the launcher's exit status.  This process never touches a GPU (counting devices does not initialise one)
    and never replaces itself (no exec).  Fewer visible GPUs than ranks: exit status 3 and no line, unless --allow-shared-device."""
    import socket
    import subprocess
    try:
        import torch
        ndev = torch.cuda.device_count()
    except Exception:
        ndev = 0
    if ndev < args.gpus and not args.allow_shared_device:
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible: refusing to report a multi-GPU number (--allow-shared-device is for tests)\n" % (args.gpus, ndev))
        return 3
    if ndev < 1:
        sys.stderr.write("bench.py needs a GPU (there is no CPU fallback of the product path)\n")
        return 1
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = 1
    for attempt in range(3):                             # (a port found free can be taken before the launcher binds it: try another one)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        got_line = False
        import threading
        err_tail = []

        def _pump():
            for raw in p.stderr:
                txt = raw.decode(errors="replace")
                err_tail.append(txt)
                del err_tail[:-200]
                sys.stderr.write(txt)
        th = threading.Thread(target=_pump, daemon=True)
        th.start()
        for raw in p.stdout:
            txt = raw.decode(errors="replace")
            is_line = False
            if txt.startswith("{"):
                try:
                    is_line = "metric" in json.loads(txt)    # (rank 0's ONE result line, nothing else that happens to look like JSON)
                except ValueError:
                    is_line = False
            if is_line:
                got_line = True
                sys.stdout.write(txt)
                sys.stdout.flush()
            else:                                        # anything else the ranks print is not the line
                sys.stderr.write(txt)
        rc = p.wait()
        th.join(timeout=5)
        if rc != 0 and not got_line and any("EADDRINUSE" in t or "Address already in use" in t or "address already in use" in t for t in err_tail):
            continue
        break
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gbases", type=float, default=0.0, help="assembly size in Gbases (0 = the full 3.16 Gbp fixture)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="strong (default for --gpus > 1: BASELINE's metric is ONE 3 Gbp assembly at 1/2/4/8 GPUs): one assembly, contigs split "
                         "over the ranks (LPT), one all-reduce per step; weak: one assembly per rank, no collective.  With --gpus > 1 the other one "
                         "is measured too (with the gather of all records to rank 0) and reported under 'second'")
    ap.add_argument("--profile", choices=("uniform", "satellite", "humanlike"), default="uniform", help="main workload (the other ones are reported under 'profiles' at N=1)")
    ap.add_argument("--assembly-index", type=int, default=0, help="seed offset of the (first) assembly: rank r of a weak run uses index + r")
    ap.add_argument("--cpu-sample-mbases", type=float, default=500.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU leg (cpu_baseline and parity)")
    ap.add_argument("--check-steps", type=int, default=-1, help="extra untimed steps whose results are digested and compared (-1: min(steps, 20); 0: none)")
    ap.add_argument("--no-profiles", action="store_true", help="skip the second workload profile")
    ap.add_argument("--no-e2e", action="store_true", help="skip the CLI end-to-end legs")
    ap.add_argument("--no-reads", action="store_true", help="skip the read-level leg (BASELINE config 5)")
    ap.add_argument("--reads-gbases", type=float, default=10.0, help="bases of FASTQ streamed through the read-level leg")
    ap.add_argument("--no-second", action="store_true", help="--gpus > 1: skip the second measurement (the other scaling mode with --gather)")
    ap.add_argument("--emulate-ranks", type=str, default="2,4,8", help="N=1: model the strong-scaling curve for these rank counts on this one GPU ('' = skip)")
    ap.add_argument("--rank-share", type=str, default="", help="N,k (profiling aid, N=1 only): the main workload is the share rank k of an N-rank strong-scaling run would own")
    ap.add_argument("--split-tol", type=float, default=0.05, help="strong scaling: contigs are cut into pieces with halos (cornetto_amd.dist.SplitPlan) when whole contigs "
                    "cannot be packed within this fraction of the fair share (the HG002 assembly: 3.8 %% over it on 8 ranks, 22 %% on 16); negative: always")
    ap.add_argument("--allow-shared-device", action="store_true", help="--gpus > 1 with fewer GPUs than ranks (tests): ranks share devices, collectives over gloo")
    ap.add_argument("--sdust-share", type=int, default=-1, help="percent of the wave slots the resident sdust waves could hold on a CU that they take while the other stream runs beside them; -1 / 0 (default): probed during warm-up (72 / 85 / 92 / 100, four untimed steps each) for every resident workload — with the coverage stage in front of the telomere scan the two host threads of a step balance at 85-92 on the uniform profile (7.6-7.8 ms per step; 72: 8.1), repeat-rich profiles want 85-100.  The rest of the slots joins in when the other thread of the step is through (cornetto_accel_boost)")
    ap.add_argument("--timing", type=int, default=1, help="HIP events in the timed steps around: 1 the three main kernels only (roofline), 2 every launch, 0 none; the extra serial pass that fills the kernel table always uses 2")
    ap.add_argument("--gather", action="store_true", help="also gather every result record to rank 0 inside the step (not part of the path: each rank owns the output of its contigs)")
    ap.add_argument("--allreduce-always", action="store_true", help="weak scaling: all-reduce the depth totals as well (treats the N assemblies as one)")
    ap.add_argument("--serial", action="store_true", help="run the stages one after the other on one stream (per-kernel timing without overlap)")
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))                     # plain `python bench.py --gpus N`: this process only starts the ranks and relays their line

    import torch
    import torch.distributed as dist
    import cornetto_amd

    R = Rank(args, torch, dist, cornetto_amd)
    rank, world = R.rank, R.world
    t_first = time.perf_counter()
    R.load(args.profile)
    if args.rank_share and world == 1:
        from cornetto_amd.dist import lpt_partition
        n_sh, k_sh = (int(x) for x in args.rank_share.split(","))
        R.wrap(lpt_partition(R.lens, n_sh)[k_sh])
        R.job_bases = R.my_bases
    t_first = time.perf_counter()
    R.step(False)                                   # the first step of a resident assembly builds its decomposition tables
    first_step_ms = (time.perf_counter() - t_first) * 1e3

    elapsed = R.timed(args.steps, args.warmup)
    elapsed_local = R.elapsed_local
    steps_ms0 = list(getattr(R, "step_ms", []))        # (of the headline measurement: later legs time steps of their own)
    serial = R.serial_kernel_times()
    n_bases = R.my_bases

    coll, shared = None, False
    if world > 1:
        coll, shared = collectives_info(R, dist, torch, elapsed_local, args.steps)
        if shared and not args.allow_shared_device:
            if rank == 0:
                sys.stderr.write("bench.py: two ranks share one device (%s): refusing to report a multi-GPU number; --allow-shared-device is for tests\n" % coll["devices"])
            R.unload()
            R.close()
            sys.exit(3)

    # ---- determinism: digest of the four result arrays of every step of an extra, untimed run ---------------
    nchk = min(args.steps, 20) if args.check_steps < 0 else args.check_steps
    det, last, gathered_digests = None, None, None
    if nchk > 0:
        digs = []
        for _ in range(nchk):
            last, gathered = R.step(False, keep=True)
            digs.append(digest(last))
            if gathered is not None and rank == 0:
                # one digest per assembly over its four gathered record arrays (global contig ids are assembly-major)
                nctg = len(R.lens)
                n_asm = world if args.scaling == "weak" else 1
                gathered_digests = []
                for a in range(n_asm):
                    part = []
                    for g in gathered:
                        sel = g[(g["ctg"] // nctg) == a].copy()
                        sel["ctg"] -= a * nctg
                        part.append(sel)
                    gathered_digests.append(digest(part))
        same = len(set(digs)) == 1
        det = {"steps": nchk, "identical": bool(same), "digest": digs[0], "what": "xxh3-128 over telomere runs, telomere windows, sdust intervals, selected coverage windows of this rank"}
        if world > 1:
            flags = [None] * world
            dist.all_gather_object(flags, bool(same))
            det["identical"] = bool(all(flags))
    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = R.job_bases / (elapsed / args.steps) / 1e9
        kavg, alg, kern = kernel_table(R, serial, n_bases)
        dom = "sdust_kernel"
        # which kernel family the library chose for this assembly (CORNETTO_SDUST_SIFT unset: by a sample of the bases)
        st0 = R.acc2.sdust_stats(R.asm2, 20, 64)
        symbol = "sd_sift" if st0 and st0.get("kernel") == "sd_sift" else "sdust_w64"
        ach = alg[dom] / (kavg[dom] * 1e-3) / 1e9 if kavg.get(dom, 0) > 0 else 0.0
        # HBM/fabric bytes per launch of the dominant kernel and its instruction mix: NOT measured in this run — taken from the
        # committed rocprofv3 PMC passes of this workload (profiles/README.md), scaled to the bases of this run
        traffic, traffic_source, traffic_ms, issue = None, None, None, None
        cands = [PMC_FILE % (rd, args.profile, symbol) for rd in PMC_ROUNDS] + [os.path.join("profiles", "r03_pmc_traffic_%s.json" % symbol), os.path.join("profiles", "r02_pmc_traffic.json")]
        for cand in cands:
            try:
                pmcs = json.load(open(os.path.join(ROOT, cand)))
                pmc = pmcs.get(symbol) or pmcs["sdust_w64"]
                traffic, traffic_source = round(pmc["hbm_bytes"] / pmc["bases"] * n_bases, 0), cand
                traffic_ms = pmc.get("production_ms") or pmcs.get("production_ms")
                break
            except Exception:
                continue
        try:
            sq_name = None
            for rd in PMC_ROUNDS:
                if os.path.exists(os.path.join(ROOT, SQ_FILE % (rd, args.profile, symbol))):
                    sq_name = SQ_FILE % (rd, args.profile, symbol)
                    break
            if sq_name is None:
                sq_name = os.path.join("profiles", "r03_sq_%s.json" % symbol)
            sq = json.load(open(os.path.join(ROOT, sq_name)))
            pl = sq["per_launch"]
            issue = {"valu_insts": pl["SQ_INSTS_VALU"], "salu_insts": pl["SQ_INSTS_SALU"], "lds_insts": pl["SQ_INSTS_LDS"],
                     "valu_busy": sq.get("valu_busy_of_kernel_time"), "per_64_bases": sq.get("per_64_bases"), "source": sq_name,
                     "counting_build_ms": sq.get("counting_build_ms", sq.get("kernel_ms")), "production_ms": sq.get("production_ms"),
                     "note": "a wave-64 vector instruction holds its SIMD's ALU for 4 cycles: valu_busy = 4 x SQ_INSTS_VALU / (SIMDs x cycles of the kernel); "
                             "not measured in this run: the committed rocprofv3 --pmc passes of the same workload"}
            # the counters describe the kernel as it was when they were collected: compare that build's time alone on the chip with this run's
            # (ms_uncontended of the dominant kernel) and say so when they are more than 10 % apart — a stale file must not survive a kernel change
            now_ms = kern.get(dom, {}).get("ms_uncontended")
            if now_ms and issue.get("production_ms"):
                issue["this_run_ms_alone"] = now_ms
                issue["stale"] = bool(abs(issue["production_ms"] - now_ms) > 0.10 * now_ms)
        except Exception:
            pass
        # ---- roofline: the dominant kernel as the contract asks (algorithmic bytes / its launch duration against the HBM peak), what really
        # bounds it, the other streaming kernels alone and inside the step, and the whole step (6 B/base: 1 B of sequence read by sdust,
        # 1 B read again by the telomere scan, 4 B of coverage)
        now_alone = kern.get(dom, {}).get("ms_uncontended")
        stale_t = bool(traffic_ms and now_alone and abs(traffic_ms - now_alone) > 0.10 * now_alone)
        kfr = {}
        for kname in ("cov_blocks", "tf_scan", dom):
            kk = kern.get(kname, {})
            ent = {"bytes_per_base": alg[kname] / n_bases}
            if kk.get("ms_uncontended"):
                g = alg[kname] / (kk["ms_uncontended"] * 1e-3) / 1e9
                ent["alone"] = {"ms": kk["ms_uncontended"], "GBps": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4)}
            if kk.get("ms"):
                g = alg[kname] / (kk["ms"] * 1e-3) / 1e9
                ent["in_step"] = {"ms": kk["ms"], "GBps": round(g, 1), "frac": round(g / HBM_PEAK_GBS, 4)}
            kfr[kname] = ent
        step_gbps = 6.0 * n_bases / (ms_per_step * 1e-3) / 1e9
        roof = {"bound": "hbm", "limited_by": "instruction issue and the LDS pipeline of the dominant kernel, not HBM (profiles/README.md)",
                "kernel": dom, "kernel_symbol": symbol, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                "traffic_ratio": round(traffic / alg[dom], 3) if traffic else None, "traffic_source_production_ms": traffic_ms, "traffic_stale": stale_t,
                "issue": issue,
                "step": {"bytes_per_base": 6, "achieved_GBps": round(step_gbps, 1), "frac": round(step_gbps / HBM_PEAK_GBS, 4),
                         "what": "all three scans of a step: 6 algorithmic bytes per base (SURVEY 8d) x bases / ms_per_step against the HBM peak"},
                "kernels": kfr,
                "note": "sdust is integer work per base, bound by instruction issue and LDS cycles, not by HBM (profiles/r05_<profile>_sq_<kernel>.json); achieved / peak / frac are "
                        "against the HBM roofline as the contract asks (1 B/base algorithmic). traffic and issue are not measured in this run: "
                        "the committed rocprofv3 PMC figures scaled by bases, flagged stale when that build's kernel time differs from this run's by more than 10 %"}
        sm_ = sorted(steps_ms0) if steps_ms0 else []
        spread = {"min": round(sm_[0], 3), "median": round(sm_[len(sm_) // 2], 3), "max": round(sm_[-1], 3), "steps": len(sm_),
                  "what": "host time from one step's return to the next's on rank 0 inside the timed region (ms_per_step is the region / steps)"} if sm_ else None
        nst = 2 if R.overlap else 1
        wl = "%s over %s synthetic HG002-like hifiasm assembly%s (%d contigs, %.3f Gbp%s, planted telomeres/STRs/N runs%s; per-base u16 depth+mq)" % (
            "telowin+sdust+noboringbits", "one" if args.scaling == "strong" or world == 1 else "%d" % world,
            "" if args.scaling == "strong" or world == 1 else " (one per GPU)", len(R.lens), sum(R.lens) / 1e9,
            " each" if args.scaling == "weak" and world > 1 else "", {"satellite": ", satellite arrays, microsatellites, poly-A", "humanlike": "; isochores 35-55 % GC, CpG depleted, 10 % Alu-like + 15 % L1-like copies, microsatellites, 3 % satellite arrays"}.get(args.profile, ""))
        line = {
            "metric": "Gbases/s scanned (telowin+sdust+boringbits)", "value": round(value, 4), "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "ms_per_step_spread": spread,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "single", "vs_baseline": None, "dtype": "u8/u16 integer",
            "data": "synthetic",
            "config": {"workload": wl, "profile": args.profile, "bases_per_gpu": n_bases, "bases_job": R.job_bases, "contigs": len(R.lens),
                       "contigs_rank0": len(R.own), "cut_contigs": (sum(1 for c in R.plan.cuts.values() if c) if getattr(R, "plan", None) is not None else 0),
                       "motif": "TTAGGG", "sdust": "-w 64 -t 20", "windows": "-w 2500 -i 50",
                       "parallelism": "contig-sharded (%s), %d process(es), 1 GPU each; per GPU %s" % (
                           "LPT over the contigs of one assembly" if args.scaling == "strong" else "one assembly per rank", world,
                           "2 HIP streams (sdust || telofind+coverage), sdust on %d %% of the wave slots" % getattr(R, "share", 100) if nst == 2 else "stages serial on one stream")},
            "roofline": roof,
            "kernels": kern,
            "stage_wall_ms": {k: round(float(np.mean(v)), 3) for k, v in R.wall.items()},
            "results_per_rank": dict(zip(("telomere_runs", "telomere_windows", "sdust_intervals", "selected_cov_windows"), R.counts)),
            "cached_across_steps": ["tile / chunk / window decomposition tables of the resident assembly and coverage (functions of the contig lengths only; built by the first "
                                    "step, %0.1f ms against %0.1f ms for a timed step)" % (first_step_ms, ms_per_step),
                                    "device workspaces and pinned result buffers (no allocation in a timed step)",
                                    "sdust: which chunks of the chunk table run long (inside repeat arrays, with other bytes) — noted by the first call for a table, "
                                    "handed out first from the second call on (a function of the resident bases, which must not change)",
                                    "telofind: the motif's match tables on the device while the motif stays the same"],
            "first_step_ms": round(first_step_ms, 3), "accel_warm_ms": round(R.warm_ms, 2),
            "sdust_share_percent": getattr(R, "share", None), "sdust_share_probe_ms": getattr(R, "share_probe_ms", None),
        }
        if coll:
            line["collectives"] = coll
        if det:
            line["determinism"] = det
        if gathered_digests:
            line["gathered_digests"] = gathered_digests
    ok = True
    if rank == 0 and world == 1 and not args.no_cpu:
        base, results = cpu_reference_leg(R.bases, R.depth, R.mq, R.offs, R.lens, R.own, int(args.cpu_sample_mbases * 1e6))
        line["cpu_baseline"] = base
        if last is None:
            last, _ = R.step(False, keep=True)
        par = check_parity(results, last, R.lo, R.hi, 0.4, 1000000, R.lens_own)
        par["against"] = "the reference's own functions (oracle/_ref)" if base["kind"] == "reference" else "the oracle port"
        line["parity"] = par
        line["parity_checked_bases"] = par["checked_bases"]
        ok = par["ok"]
        del results
    if det and not det["identical"]:
        ok = False
    last = None
    if rank == 0 and world == 1 and args.emulate_ranks.strip():
        ns = [int(x) for x in args.emulate_ranks.split(",") if x.strip()]
        line["scaling_model"] = emulate_ranks(R, ns, max(3, min(args.steps, 10)), line["ms_per_step"])
    if world > 1 and not args.no_second:
        # the other scaling mode, with the gather of every record to rank 0: BASELINE config 4 is "8 assemblies, RCCL gather of BED intervals"
        other = "weak" if args.scaling == "strong" else "strong"
        st2 = max(3, min(args.steps, 10))
        el2 = measure(R, args, other, args.profile, st2, 1, True)
        if rank == 0:
            line["second"] = {"scaling": other, "gather": True, "steps": st2, "ms_per_step": round(el2 / st2 * 1e3, 3),
                              "value": round(R.job_bases / (el2 / st2) / 1e9, 4), "unit": "Gbases/s", "bases_job": R.job_bases,
                              "stage_wall_ms": {k: round(float(np.mean(v)), 3) for k, v in R.wall.items()}}
    if rank == 0 and world == 1 and getattr(R, "plan", None) is None:
        # the cold pass once more, in a process that is warm: the same resident inputs wrapped as new objects (no decomposition tables, no result
        # counts to size anything by) — what a second assembly costs a panel run that scans eight (BASELINE config 4); first_step_ms above is
        # the first step of the process (code objects, workspaces and pinned pools come into being in it)
        R.wrap(R.own)
        keep_wall, R.wall = R.wall, {}
        t_c = time.perf_counter()
        R.step(True)
        line["cold_step_ms"] = line["first_step_ms"]
        line["cold_step_ms_next_assembly"] = round((time.perf_counter() - t_c) * 1e3, 3)
        line["cold_step_next_assembly_stage_ms"] = {k: round(v[0], 3) for k, v in R.wall.items()}
        R.wall = keep_wall
    if rank == 0 and world == 1 and not args.no_e2e:
        line["e2e"] = e2e_cli(R, cornetto_amd)
    if rank == 0 and world == 1 and not args.no_profiles:
        if R.scaling != args.scaling or R.profile != args.profile:
            R.unload()
            R.load(args.profile, args.scaling)
        profs = {args.profile: {"ms_per_step": line["ms_per_step"], "gbases_s": line["value"],
                                "sdust_kernel_ms": line["kernels"].get("sdust_kernel", {}).get("ms"),
                                "sdust_kernel_ms_uncontended": line["kernels"].get("sdust_kernel", {}).get("ms_uncontended"),
                                "results": line["results_per_rank"]}}
        st = R.acc2.sdust_stats(R.asm2, 20, 64) if hasattr(R.acc2, "sdust_stats") else None
        if st:
            profs[args.profile]["sdust_stats"] = st
        for other in ("uniform", "satellite", "humanlike"):
            if other == args.profile:
                continue
            R.unload()
            R.load(other)
            profs[other] = profile_leg(R, min(args.steps, 5), parity=not args.no_cpu)
            if not profs[other].get("parity", {}).get("ok", True):
                ok = False
        line["profiles"] = profs
    R.unload()
    if rank == 0 and world == 1 and not args.no_e2e:
        line["e2e"]["noboringbits"] = e2e_noboringbits(R, torch, cornetto_amd)
        e2 = line["e2e"]
        e2_ok = [e2.get("sdust", {}).get("stdout_equals_step_records", True), e2.get("telofind", {}).get("stdout_equals_step_records", True),
                 e2.get("same_as_reference", {}).get("ok", True), e2["noboringbits"].get("same_as_reference", {}).get("ok", True),
                 e2["noboringbits"].get("sharded_ingest_same_gpu_twice", {}).get("same_stdout", True)]
        e2["parity_ok"] = bool(all(e2_ok))
        if not e2["parity_ok"]:
            ok = False
    if not args.no_reads:                               # every rank: N > 1 streams one FASTQ stream sharded by cumulative bases (config 5)
        rd = reads_leg(R, torch, dist, cornetto_amd, args.reads_gbases)
        if rank == 0:
            line["reads"] = rd
            if rd.get("parity") is not None and not rd["parity"].get("ok", True):
                ok = False
    if rank == 0:
        print(json.dumps(line), flush=True)
    R.close()
    if not ok:
        sys.stderr.write("bench.py: parity or determinism check FAILED (see the JSON line)\n")
        sys.exit(1)


if __name__ == "__main__":
    main()
