"""cornetto_amd — MI355X-native implementation of cornetto's panel-creation hot path.

The product is native: ``libcornetto_hip.so`` (hand-written HIP kernels for gfx950 behind the C ABI of
``include/cornetto_accel.h``) and the C host CLI ``cornetto`` (``cornetto_amd/cli``).  This module is a thin
ctypes binding of that C ABI, used by the tests, ``bench.py`` and the multi-process drivers: plumbing, not
a second implementation.  There is no Python or CPU fallback — if the shared library is missing, import
of :func:`lib` fails loudly.
"""
import ctypes as C
import os

import numpy as np

__all__ = ["lib", "Accel", "AccelError", "HIT_DT", "WIN_DT", "IVL_DT", "TELROW_DT", "khash_str_order", "panel_boring", "REG_DT", "REGREC_DT", "build",
           "LIB_PATH", "CLI_PATH"]

HERE = os.path.dirname(os.path.abspath(__file__))
DEV_LIB_PATH = os.path.join(HERE, "libcornetto_hip_dev.so")   # the same sources with -DCN_DEV: the development switches (CORNETTO_SDUST_CHUNK, ...) exist in this build only
LIB_PATH = os.environ.get("CORNETTO_LIB") or os.path.join(HERE, "libcornetto_hip.so")    # (CORNETTO_LIB: another build of the same library — A/B runs of kernel variants on one box)
CLI_PATH = os.path.join(HERE, "cornetto")

HIT_DT = np.dtype([("ctg", "<i4"), ("strand", "<i4"), ("start", "<i4"), ("end", "<i4")])
WIN_DT = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("end", "<i4"), ("car", "<i4")])
IVL_DT = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("finish", "<i4")])
TELROW_DT = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("end", "<i4"), ("matched", "<i4")])
FQREC_DT = np.dtype([("head", "<i8"), ("seq", "<i8"), ("qual", "<i8"), ("len", "<i4"), ("name_len", "<i4"),
                     ("comment_len", "<i4"), ("keep", "<i4")])
FAREC_DT = np.dtype([("head", "<i8"), ("len", "<i8"), ("name_len", "<i4"), ("pad", "<i4")])
REG_DT = np.dtype([("st", "<i4"), ("end", "<i4"), ("depth", "<i4"), ("mq_depth", "<i4")])
REGREC_DT = np.dtype([("ctg", "<i4"), ("st", "<i4"), ("end", "<i4"), ("depth", "<i4"), ("mq_depth", "<i4")])
REGPK_DT = np.dtype([("st", "<i4"), ("depth", "<u2"), ("mq_depth", "<u2")])


class BgErr(C.Structure):
    _fields_ = [("kind", C.c_int32), ("record", C.c_int64), ("a", C.c_int32), ("b", C.c_int32)]


class StepOpt(C.Structure):
    """cornetto_step_opt_t (include/cornetto_accel.h)"""
    _fields_ = [("motif", C.c_char_p), ("thr_adj", C.c_double), ("window_size", C.c_int32), ("window_inc", C.c_int32), ("low_cov", C.c_float),
                ("high_cov", C.c_float), ("low_mq", C.c_float), ("edge_len", C.c_int32), ("min_ctg_len", C.c_int32), ("boring", C.c_int32)]


SUMS_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint64), C.c_void_p)


class BedgraphFormatError(ValueError):
    """a check of the reference's get_depths() failed (kind / record / numbers as in cornetto_bgerr_t)"""

    def __init__(self, kind, record, a, b):
        super().__init__("bedgraph check %d failed at record %d (%d, %d)" % (kind, record, a, b))
        self.kind, self.record, self.a, self.b = kind, record, a, b


class AccelError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("cornetto_accel status %d: %s" % (status, msg))
        self.status = status


def build(verbose=False):
    """compile libcornetto_hip.so and the CLI in-tree (hipcc --offload-arch=gfx950)"""
    import subprocess
    subprocess.check_call(["make", "-C", HERE] + ([] if verbose else ["-s"]))


_libs = {}


def lib(dev=False):
    """the loaded C ABI; raises if the HIP library has not been built.  dev=True: the development build of the same sources
    (libcornetto_hip_dev.so, -DCN_DEV), the only one that reads the development switches from the environment — a second, independent copy
    of the library in the process (own handles, own result pool)"""
    path = DEV_LIB_PATH if dev else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `make -C cornetto_amd` (or __graft_entry__.build()); "
                          "there is no fallback implementation" % path)
    L = C.CDLL(path)
    vp, i32, i64, cp = C.c_void_p, C.c_int32, C.c_int64, C.c_char_p
    pp = C.POINTER(vp)
    sig = {
        "cornetto_accel_device_count": (C.c_int, []),
        "cornetto_accel_open": (C.c_int, [pp, C.c_int, vp]),
        "cornetto_accel_close": (None, [vp]),
        "cornetto_accel_last_error": (cp, [vp]),
        "cornetto_accel_strerror": (cp, [C.c_int]),
        "cornetto_free": (None, [vp]),
        "cornetto_accel_last_timing": (C.c_int, [vp, C.POINTER(cp), C.POINTER(C.c_float), C.c_int]),
        "cornetto_accel_set_share": (C.c_int, [vp, C.c_int]),
        "cornetto_accel_boost": (C.c_int, [vp, C.c_int]),
        "cornetto_accel_launch_count": (C.c_uint64, [vp]),
        "cornetto_accel_set_lazy": (C.c_int, [vp, C.c_int]),
        "cornetto_accel_wait": (C.c_int, [vp]),
        "cornetto_accel_set_timing": (C.c_int, [vp, C.c_int]),
        "cornetto_accel_warm": (C.c_int, [vp, C.c_int]),
        "cornetto_accel_sdust_stats": (C.c_int, [vp, C.c_int, vp, C.c_int]),
        "cornetto_cov_select_merged": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int, C.c_int32, C.c_int32, C.POINTER(vp), C.POINTER(C.c_int64)]),
        "cornetto_ivl_merge": (C.c_int, [vp, vp, C.c_int64, C.c_int32, C.POINTER(vp), C.POINTER(C.c_int64)]),
        "cornetto_panel_defaults": (None, [vp]),
        "cornetto_panel_defaults_recreate": (None, [vp]),
        "cornetto_panel_boring": (C.c_int, [vp, C.c_int32, vp, C.c_int64, vp, C.c_int64, vp, C.POINTER(vp), C.POINTER(C.c_int64)]),
        "cornetto_telobreaks": (C.c_int, [vp, vp, C.c_int32, vp, C.c_int64, vp, C.c_int64, C.POINTER(vp), C.POINTER(C.c_int64)]),
        "cornetto_khash_str_order": (C.c_int32, [C.POINTER(C.c_char_p), C.c_int32, vp, vp]),
        "cornetto_fastq_split": (C.c_int, [vp, vp, i64, C.c_int, i32, pp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), pp]),
        "cornetto_fasta_split": (C.c_int, [vp, vp, i64, C.c_int, pp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), pp]),
        "cornetto_text_open": (C.c_int, [vp, i64, pp]),
        "cornetto_text_free": (None, [vp, vp]),
        "cornetto_text_put": (C.c_int, [vp, vp, vp, i64, i64, C.c_int]),
        "cornetto_text_wait": (C.c_int, [vp, vp, C.c_int]),
        "cornetto_fasta_split_text": (C.c_int, [vp, vp, i64, C.c_int, pp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), pp]),
        "cornetto_asm_upload": (C.c_int, [vp, vp, vp, i32, pp]),
        "cornetto_asm_wrap": (C.c_int, [vp, vp, vp, vp, i32, pp]),
        "cornetto_asm_free": (None, [vp, vp]),
        "cornetto_telofind": (C.c_int, [vp, vp, cp, pp, C.POINTER(i64)]),
        "cornetto_telowin_threshold": (C.c_double, [C.c_double, C.c_double]),
        "cornetto_telowin": (C.c_int, [vp, vp, i64, vp, i32, C.c_double, pp, C.POINTER(i64)]),
        "cornetto_telo_scan": (C.c_int, [vp, vp, cp, C.c_double, pp, C.POINTER(i64), pp, C.POINTER(i64)]),
        "cornetto_sdust_asm": (C.c_int, [vp, vp, i32, i32, pp, C.POINTER(i64)]),
        "cornetto_sdust_asm_begin": (C.c_int, [vp, vp, i32, i32]),
        "cornetto_sdust_asm_end": (C.c_int, [vp, vp, i32, i32, pp, C.POINTER(i64)]),
        "cornetto_sdust": (C.POINTER(C.c_uint64), [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
        "cornetto_sdust_buf_init": (vp, [vp]),
        "cornetto_sdust_buf_destroy": (None, [vp]),
        "cornetto_sdust_core": (C.POINTER(C.c_uint64), [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), vp]),
        "cornetto_cov_upload": (C.c_int, [vp, vp, vp, vp, i32, pp]),
        "cornetto_cov_wrap": (C.c_int, [vp, vp, vp, vp, vp, i32, pp]),
        "cornetto_cov_free": (None, [vp, vp]),
        "cornetto_cov_shard": (C.c_int, [vp, vp, vp, vp, i32, pp]),
        "cornetto_n_reg": (i32, [i32, i32, i32]),
        "cornetto_regs_assert": (i32, [i32, i32, i32]),
        "cornetto_cov_prepare": (C.c_int, [vp, vp, i32, i32, C.POINTER(C.c_uint64)]),
        "cornetto_cov_regs": (C.c_int, [vp, vp, i32, vp]),
        "cornetto_cov_threshold": (i32, [C.c_float, i32]),
        "cornetto_cov_select": (C.c_int, [vp, vp, i32, i32, C.c_float, i32, i32, C.c_int, pp, C.POINTER(i64)]),
        "cornetto_cov_select_packed": (C.c_int, [vp, vp, i32, i32, C.c_float, i32, i32, C.c_int, pp, C.POINTER(i64), pp]),
        "cornetto_panel_step": (C.c_int, [vp, vp, vp, C.POINTER(StepOpt), vp, vp, C.POINTER(C.c_uint64), C.POINTER(i32), pp, C.POINTER(i64), pp, pp, C.POINTER(i64),
                                          pp, C.POINTER(i64)]),
        "cornetto_cov_n": (i32, [vp]),
        "cornetto_cov_lens": (C.POINTER(i32), [vp]),
        "cornetto_pinned_alloc": (vp, [C.c_size_t]),
        "cornetto_pinned_free": (None, [vp]),
        "cornetto_bgin_open": (C.c_int, [vp, pp]),
        "cornetto_bgin_close": (None, [vp, vp]),
        "cornetto_bgin_feed": (C.c_int, [vp, vp, cp, i64, cp, i64, C.c_int]),
        "cornetto_bgin_prefetch": (C.c_int, [vp, vp, cp, i64, cp, i64]),
        "cornetto_bgin_pending": (None, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "cornetto_bgin_unmatched_mq": (i64, [vp]),
        "cornetto_bgin_error": (C.POINTER(BgErr), [vp]),
        "cornetto_bgin_done": (C.c_int, [vp]),
        "cornetto_bgin_finish": (C.c_int, [vp, vp, pp, C.POINTER(i32), C.POINTER(C.POINTER(cp)), C.POINTER(i64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    L._declared = sorted(sig)
    _libs[path] = L
    return L


class _Owner:
    """keeps a library-owned result buffer alive for the numpy view over it; released with cornetto_free()"""

    def __init__(self, L, ptr):
        self.L, self.ptr = L, ptr

    def __del__(self):
        try:
            if self.ptr:
                self.L.cornetto_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def _take(L, ptr, n, dt):
    """zero-copy numpy view of a result array owned by library L (freed when the array is garbage collected)"""
    if not n:
        L.cornetto_free(ptr)
        return np.zeros(0, dtype=dt)
    addr = ptr.value if isinstance(ptr, C.c_void_p) else int(ptr)
    buf = (C.c_char * (n * dt.itemsize)).from_address(addr)
    buf._owner = _Owner(L, addr)          # the ctypes array is the numpy base; the owner dies with it
    return np.frombuffer(buf, dtype=dt)


class _Resident:
    def __init__(self, acc, ptr, free, lens):
        self.acc, self.ptr, self._free, self.lens = acc, ptr, free, list(lens)

    def close(self):
        if self.ptr is not None:
            self._free(self.acc.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Accel:
    """one device + stream; mirrors the handle of include/cornetto_accel.h"""

    def __init__(self, device=0, stream=None, dev=False):
        self.L = lib(dev)
        h = C.c_void_p()
        rc = self.L.cornetto_accel_open(C.byref(h), device, stream)
        if rc != 0:
            raise AccelError(rc, self.L.cornetto_accel_strerror(rc).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.cornetto_accel_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise AccelError(rc, self.L.cornetto_accel_last_error(self.h).decode() or
                             self.L.cornetto_accel_strerror(rc).decode())

    def launch_count(self):
        """launches of resident sdust waves on this handle so far: cornetto_accel_launch_count()"""
        return int(self.L.cornetto_accel_launch_count(self.h))

    def set_lazy(self, on=True):
        """large result copies on a stream of their own; the arrays the calls return hold their content after wait(): cornetto_accel_set_lazy()"""
        self._chk(self.L.cornetto_accel_set_lazy(self.h, 1 if on else 0))

    def wait(self):
        self._chk(self.L.cornetto_accel_wait(self.h))

    def boost(self, on=True):
        """the rest of the device is free (on) / in use again (off): see cornetto_accel_boost(); callable from another thread"""
        self.L.cornetto_accel_boost(self.h, 1 if on else 0)

    def set_share(self, percent):
        """percent of every CU the resident sdust kernel may occupy (another handle computes beside this one)"""
        self._chk(self.L.cornetto_accel_set_share(self.h, int(percent)))

    def set_timing(self, level):
        self._chk(self.L.cornetto_accel_set_timing(self.h, level))

    def last_timing(self):
        """[(kernel name, ms)] of the most recent compute call (HIP events on the handle's stream)"""
        names = (C.c_char_p * 64)()
        ms = (C.c_float * 64)()
        n = self.L.cornetto_accel_last_timing(self.h, names, ms, 64)
        return [(names[i].decode(), float(ms[i])) for i in range(min(n, 64))]

    # ---- sequences -------------------------------------------------------------------------------
    def asm_upload(self, seqs):
        """seqs: list of bytes / uint8 arrays -> resident assembly"""
        arrs = [np.frombuffer(bytes(s), dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else
                np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        lens = np.array([a.size for a in arrs], dtype=np.int64)
        out = C.c_void_p()
        self._chk(self.L.cornetto_asm_upload(self.h, ptrs, lens.ctypes.data, n, C.byref(out)))
        return _Resident(self, out, self.L.cornetto_asm_free, lens)

    def asm_wrap(self, dev_ptr, offsets, lens):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        out = C.c_void_p()
        self._chk(self.L.cornetto_asm_wrap(self.h, dev_ptr, offsets.ctypes.data, lens.ctypes.data, len(lens), C.byref(out)))
        return _Resident(self, out, self.L.cornetto_asm_free, lens)

    def fastq_split(self, text, final=True, min_len=0, want_reads=False):
        """text: bytes-like FASTQ piece (or (address, size) of e.g. pinned memory) -> (records FQREC_DT, consumed bytes,
        plain flag, resident reads or None): see cornetto_fastq_split() in include/cornetto_accel.h"""
        if isinstance(text, tuple):
            addr, n = text
            keep = None
        else:
            keep = np.frombuffer(bytes(text), dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
            addr, n = keep.ctypes.data, keep.size
        p, cnt, used, plain, reads = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_void_p()
        self._chk(self.L.cornetto_fastq_split(self.h, addr, n, 1 if final else 0, min_len, C.byref(p), C.byref(cnt), C.byref(used),
                                              C.byref(plain), C.byref(reads) if want_reads else None))
        recs = _take(self.L, p, cnt.value, FQREC_DT)
        res = _Resident(self, reads, self.L.cornetto_asm_free, recs["len"][recs["keep"] == 1]) if want_reads else None
        return recs, used.value, bool(plain.value), res

    def warm(self, what=7):
        """cornetto_accel_warm(): every entry point of the groups (1 sdust, 2 telo, 4 coverage) once on a built-in 4 kb input"""
        self._chk(self.L.cornetto_accel_warm(self.h, what))

    def fasta_split(self, text, final=True, want_seqs=False):
        """text: bytes-like FASTA piece beginning with '>' (or (address, size)) -> (records FAREC_DT, consumed bytes, plain
        flag, resident sequences or None): see cornetto_fasta_split() in include/cornetto_accel.h"""
        if isinstance(text, tuple):
            addr, n = text
            keep = None
        else:
            keep = np.frombuffer(bytes(text), dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
            addr, n = keep.ctypes.data, keep.size
        p, cnt, used, plain, seqs = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_void_p()
        self._chk(self.L.cornetto_fasta_split(self.h, addr, n, 1 if final else 0, C.byref(p), C.byref(cnt), C.byref(used), C.byref(plain),
                                              C.byref(seqs) if want_seqs else None))
        recs = _take(self.L, p, cnt.value, FAREC_DT)
        res = _Resident(self, seqs, self.L.cornetto_asm_free, recs["len"]) if want_seqs else None
        return recs, used.value, bool(plain.value), res

    def fasta_split_slabs(self, text, slab_bytes, final=True, want_seqs=False):
        """the same through cornetto_text_open / _put / cornetto_fasta_split_text: `text` reaches the device in slabs of slab_bytes from a ring of
        four pinned slabs, out of order across the two copy queues"""
        keep = np.frombuffer(bytes(text), dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
        n = keep.size
        t = C.c_void_p()
        self._chk(self.L.cornetto_text_open(self.h, max(1, n), C.byref(t)))
        ring = [self.L.cornetto_pinned_alloc(slab_bytes) for _ in range(4)]
        try:
            offs = list(range(0, n, slab_bytes))
            order = offs[1::2] + offs[0::2]                  # (the device offsets are the caller's: any order)
            for k, at in enumerate(order):
                slot = k & 3
                if k >= 4:
                    self._chk(self.L.cornetto_text_wait(self.h, t, slot))
                m = min(slab_bytes, n - at)
                C.memmove(ring[slot], keep.ctypes.data + at, m)
                self._chk(self.L.cornetto_text_put(self.h, t, ring[slot], m, at, slot))
            p, cnt, used, plain, seqs = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_void_p()
            self._chk(self.L.cornetto_fasta_split_text(self.h, t, n, 1 if final else 0, C.byref(p), C.byref(cnt), C.byref(used), C.byref(plain),
                                                       C.byref(seqs) if want_seqs else None))
        finally:
            self.L.cornetto_text_free(self.h, t)
            for r in ring:
                self.L.cornetto_pinned_free(r)
        recs = _take(self.L, p, cnt.value, FAREC_DT)
        res = _Resident(self, seqs, self.L.cornetto_asm_free, recs["len"]) if want_seqs else None
        return recs, used.value, bool(plain.value), res

    # ---- telofind / telowin ----------------------------------------------------------------------
    def telofind(self, asm, motif=b"TTAGGG"):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.cornetto_telofind(self.h, asm.ptr, motif, C.byref(p), C.byref(n)))
        return _take(self.L, p, n.value, HIT_DT)

    def telowin_threshold(self, thr, identity):
        return self.L.cornetto_telowin_threshold(thr, identity)

    def telowin(self, hits, ctg_len, thr_adj):
        hits = np.ascontiguousarray(hits, dtype=HIT_DT)
        ctg_len = np.ascontiguousarray(ctg_len, dtype=np.int32)
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.cornetto_telowin(self.h, hits.ctypes.data, hits.size, ctg_len.ctypes.data, ctg_len.size,
                                          thr_adj, C.byref(p), C.byref(n)))
        return _take(self.L, p, n.value, WIN_DT)

    def telo_scan(self, asm, motif, thr_adj, want_hits=True):
        ph, nh, pw, nw = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        self._chk(self.L.cornetto_telo_scan(self.h, asm.ptr, motif, thr_adj,
                                            C.byref(ph) if want_hits else None, C.byref(nh) if want_hits else None,
                                            C.byref(pw), C.byref(nw)))
        hits = _take(self.L, ph, nh.value, HIT_DT) if want_hits else None
        return hits, _take(self.L, pw, nw.value, WIN_DT)

    # ---- sdust -----------------------------------------------------------------------------------
    def sdust(self, asm, T=20, W=64):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.cornetto_sdust_asm(self.h, asm.ptr, T, W, C.byref(p), C.byref(n)))
        return _take(self.L, p, n.value, IVL_DT)

    def sdust_begin(self, asm, T=20, W=64):
        """cornetto_sdust_asm_begin(): queue the call without waiting where the last call's counts allow it (else nothing); sdust_end() delivers"""
        self._chk(self.L.cornetto_sdust_asm_begin(self.h, asm.ptr, T, W))

    def sdust_end(self, asm, T=20, W=64):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.cornetto_sdust_asm_end(self.h, asm.ptr, T, W, C.byref(p), C.byref(n)))
        return _take(self.L, p, n.value, IVL_DT)

    def sdust_stats(self, asm, T=20, W=64):
        """one extra sdust pass with the counting build of the kernel -> its counters as a dict (see
        cornetto_accel_sdust_stats() in include/cornetto_accel.h); None for parameters the production kernel does not take"""
        self.L.cornetto_accel_sdust_stats(self.h, 1, None, 0)
        try:
            self.sdust(asm, T, W)
        finally:
            out = np.zeros(256, dtype=np.uint64)
            n = self.L.cornetto_accel_sdust_stats(self.h, 0, out.ctypes.data, 256)
        s = [int(x) for x in out[:n]]
        waves = s[254]
        if n > 207 and s[206]:
            # the sift / resolve kernel (csrc/sdust_sift.hpp): positions after each filter, steps of the stepping stages
            return {"kernel": "sd_sift", "chunks": s[255], "tiles_of_64_bases": s[206], "positions_ct_above_T10": s[203], "after_L1": s[204], "after_L2": s[205],
                    "resolve_steps": s[200], "resolve_window_reads": s[201], "dp_tiles": s[208], "passes_with_candidates": s[202], "base_by_base_steps_of_chunks_with_other_bytes": s[207],
                    "window_reads_behind_a_gap_of": {"2": s[209], "3-4": s[210], "5-8": s[211]} if n > 211 else None}
        if not waves or not s[2]:
            return None
        hist = []
        for b in range(32):
            w = s[16 + 4 * b]
            if w:
                hist.append({"ms": [b * 0.5, b * 0.5 + 0.5], "waves": w, "jobs": s[18 + 4 * b], "find_perfect": s[19 + 4 * b], "find_perfect_with_candidates": s[17 + 4 * b]})
        return {"waves": waves, "chunks": s[255], "chunks_sampled_low_complexity": s[7], "wave_steps": s[2], "find_perfect_calls": s[3],
                "find_perfect_with_candidates": s[10], "plain_groups": s[4], "wave_ms_avg": round(s[5] / waves / 1e5, 3), "wave_ms_max": round(s[6] / 1e5, 3),
                "queue_fetch_rounds_per_wave": round(s[11] / waves, 1), "wave_time_histogram": hist}

    # ---- panel interval stage ---------------------------------------------------------------------
    def cov_select_merged(self, cov, lo, hi, low_mq, edge_len, min_ctg_len, boring, merge_dist=1000, min_len=30000):
        """cov_select + bedtools merge -d merge_dist + length filter, all on the device -> IVL_DT rows"""
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.cornetto_cov_select_merged(self.h, cov.ptr, lo, hi, low_mq, edge_len, min_ctg_len, 1 if boring else 0, merge_dist, min_len,
                                                    C.byref(p), C.byref(n)))
        return _take(self.L, p, n.value, IVL_DT)

    def ivl_merge(self, ivls, dist):
        ivls = np.ascontiguousarray(ivls, dtype=IVL_DT)
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.cornetto_ivl_merge(self.h, ivls.ctypes.data, len(ivls), dist, C.byref(p), C.byref(n)))
        return _take(self.L, p, n.value, IVL_DT)

    # ---- telobreaks ------------------------------------------------------------------------------
    def telobreaks(self, ctg_len, sd, tel):
        """ctg_len: int32 per contig; sd: IVL_DT rows (ctg, start, finish); tel: TELROW_DT rows -> IVL_DT rows
        {ctg, first - 1 clamped at 0, last} of the low-complexity runs holding a telomere row with its flanks"""
        ctg_len = np.ascontiguousarray(ctg_len, dtype=np.int32)
        sd = np.ascontiguousarray(sd, dtype=IVL_DT)
        tel = np.ascontiguousarray(tel, dtype=TELROW_DT)
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.cornetto_telobreaks(self.h, ctg_len.ctypes.data, len(ctg_len), sd.ctypes.data, len(sd), tel.ctypes.data, len(tel),
                                             C.byref(p), C.byref(n)))
        return _take(self.L, p, n.value, IVL_DT)

    # ---- coverage --------------------------------------------------------------------------------
    def cov_upload(self, depths, mqs):
        d = [np.ascontiguousarray(x, dtype=np.uint16) for x in depths]
        q = [np.ascontiguousarray(x, dtype=np.uint16) for x in mqs]
        n = len(d)
        pd = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in d])
        pq = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in q])
        lens = np.array([a.size for a in d], dtype=np.int32)
        out = C.c_void_p()
        self._chk(self.L.cornetto_cov_upload(self.h, pd, pq, lens.ctypes.data, n, C.byref(out)))
        return _Resident(self, out, self.L.cornetto_cov_free, lens)

    def cov_wrap(self, d_depth, d_mq, offsets, lens):
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        out = C.c_void_p()
        self._chk(self.L.cornetto_cov_wrap(self.h, d_depth, d_mq, offsets.ctypes.data, lens.ctypes.data, len(lens), C.byref(out)))
        return _Resident(self, out, self.L.cornetto_cov_free, lens)

    def cov_shard(self, src_acc, cov, ctgs):
        """contigs `ctgs` of `cov` (resident on src_acc's device) as a new coverage object on this handle's device"""
        ctgs = np.ascontiguousarray(ctgs, dtype=np.int32)
        out = C.c_void_p()
        self._chk(self.L.cornetto_cov_shard(src_acc.h, cov.ptr, self.h, ctgs.ctypes.data, len(ctgs), C.byref(out)))
        return _Resident(self, out, self.L.cornetto_cov_free, [cov.lens[i] for i in ctgs])

    def cov_prepare(self, cov, w=2500, inc=50):
        sums = (C.c_uint64 * 3)()
        self._chk(self.L.cornetto_cov_prepare(self.h, cov.ptr, w, inc, sums))
        cov.w, cov.inc = w, inc
        return int(sums[0]), int(sums[1]), int(sums[2])

    def cov_regs(self, cov, ctg):
        n = self.L.cornetto_n_reg(int(cov.lens[ctg]), cov.w, cov.inc)
        out = np.zeros(n, dtype=REG_DT)
        self._chk(self.L.cornetto_cov_regs(self.h, cov.ptr, ctg, out.ctypes.data))
        return out

    def cov_threshold(self, factor, mean):
        return self.L.cornetto_cov_threshold(factor, mean)

    def cov_select(self, cov, lo, hi, low_mq, edge_len, min_ctg_len, boring):
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.cornetto_cov_select(self.h, cov.ptr, lo, hi, low_mq, edge_len, min_ctg_len, int(boring),
                                             C.byref(p), C.byref(n)))
        return _take(self.L, p, n.value, REGREC_DT)

    def cov_select_packed(self, cov, lo, hi, low_mq, edge_len, min_ctg_len, boring):
        """-> (REGPK_DT records {st, depth, mq_depth}, int64 ctg_first[n + 1]): the windows of contig i are
        recs[ctg_first[i]:ctg_first[i + 1]], end = min(st + w, len)"""
        p, n, cf = C.c_void_p(), C.c_int64(), C.c_void_p()
        self._chk(self.L.cornetto_cov_select_packed(self.h, cov.ptr, lo, hi, low_mq, edge_len, min_ctg_len, int(boring),
                                                    C.byref(p), C.byref(n), C.byref(cf)))
        return _take(self.L, p, n.value, REGPK_DT), _take(self.L, cf, len(cov.lens) + 1, np.dtype("<i8"))

    def panel_step(self, asm, cov, motif, thr_adj, w=2500, inc=50, low_cov=0.4, high_cov=2.5, low_mq=0.4, edge_len=100000, min_ctg_len=1000000, boring=False,
                   exchange=None):
        """cornetto_panel_step(): cov_prepare -> thresholds -> cov_select_packed -> telo_scan in one call with two synchronisations.
        exchange(sums) -> sums (three ints): the all-reduce over the ranks, or None.
        -> (sums, (lo, hi), packed records, ctg_first, hits, windows)"""
        opt = StepOpt(motif, thr_adj, w, inc, low_cov, high_cov, low_mq, edge_len, min_ctg_len, int(bool(boring)))
        box = {}

        def _x(ptr, _ctx):
            try:
                r = exchange((int(ptr[0]), int(ptr[1]), int(ptr[2])))
                ptr[0], ptr[1], ptr[2] = int(r[0]), int(r[1]), int(r[2])
                return 0
            except BaseException as e:      # (an exception must not travel through the C frames)
                box["err"] = e
                return 1
        cb = SUMS_FN(_x) if exchange is not None else None
        cb_p = C.cast(cb, C.c_void_p) if cb is not None else None
        sums = (C.c_uint64 * 3)()
        thr = (C.c_int32 * 2)()
        p, n, cf, ph, nh, pw, nw = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        rc = self.L.cornetto_panel_step(self.h, asm.ptr, cov.ptr, C.byref(opt), cb_p, None, sums, thr, C.byref(p), C.byref(n), C.byref(cf), C.byref(ph), C.byref(nh),
                                        C.byref(pw), C.byref(nw))
        if "err" in box:
            raise box["err"]
        self._chk(rc)
        cov.w, cov.inc = w, inc
        return ((int(sums[0]), int(sums[1]), int(sums[2])), (int(thr[0]), int(thr[1])), _take(self.L, p, n.value, REGPK_DT), _take(self.L, cf, len(cov.lens) + 1, np.dtype("<i8")),
                _take(self.L, ph, nh.value, HIT_DT), _take(self.L, pw, nw.value, WIN_DT))

    @staticmethod
    def unpack_regs(recs, ctg_first, lens, w):
        """packed selection -> REGREC_DT rows (ctg, st, end, depth, mq_depth)"""
        out = np.zeros(len(recs), dtype=REGREC_DT)
        counts = np.diff(np.asarray(ctg_first, dtype=np.int64))
        ctg = np.repeat(np.arange(len(counts), dtype=np.int32), counts)
        out["ctg"] = ctg
        out["st"] = recs["st"]
        out["end"] = np.minimum(recs["st"].astype(np.int64) + w, np.asarray(lens, dtype=np.int64)[ctg]).astype(np.int32)
        out["depth"] = recs["depth"]
        out["mq_depth"] = recs["mq_depth"]
        return out

    # ---- bedgraph ingest ---------------------------------------------------------------------------
    def bedgraph_ingest(self, tot_pieces, mq_pieces, prefetch=0):
        """stream the two per-base bedgraphs (iterables of bytes pieces, any split points) through the device
        parser; returns (resident coverage, [contig names], clamped count).  Raises BedgraphFormatError.
        prefetch: 1 = cornetto_bgin_prefetch() of the next pieces in front of every feed (two pieces on their way, as the CLI does);
        2 = the same, and every third prefetch is of pieces that are NOT fed next (the feed must ignore what is staged)"""
        L = self.L
        bg = C.c_void_p()
        self._chk(L.cornetto_bgin_open(self.h, C.byref(bg)))
        try:
            ta, qa = list(tot_pieces), list(mq_pieces)
            n_t, n_q = len(ta), len(qa)
            n = max(n_t, n_q, 1)
            # (ctypes hands a bytes object's own buffer to a c_char_p parameter: the same address for the prefetch and the feed that follows)
            ta += [b""] * (n - n_t)
            qa += [b""] * (n - n_q)
            if prefetch:
                self._chk(L.cornetto_bgin_prefetch(self.h, bg, ta[0], len(ta[0]), qa[0], len(qa[0])))
            for i in range(n):
                t, q = ta[i], qa[i]
                fin = (1 if i >= n_t - 1 else 0) | (2 if i >= n_q - 1 else 0)
                if prefetch and i + 1 < n:
                    j = i + 1 if not (prefetch == 2 and i % 3 == 2) else 0          # (mode 2: something else than the next pieces now and then)
                    self._chk(L.cornetto_bgin_prefetch(self.h, bg, ta[j], len(ta[j]), qa[j], len(qa[j])))
                rc = L.cornetto_bgin_feed(self.h, bg, t, len(t), q, len(q), fin)
                if rc == -6:
                    e = L.cornetto_bgin_error(bg).contents
                    raise BedgraphFormatError(e.kind, e.record, e.a, e.b)
                self._chk(rc)
                if L.cornetto_bgin_done(bg):
                    break
            cov, nc, names, ncl = C.c_void_p(), C.c_int32(), C.POINTER(C.c_char_p)(), C.c_int64()
            self._chk(L.cornetto_bgin_finish(self.h, bg, C.byref(cov), C.byref(nc), C.byref(names), C.byref(ncl)))
            nm = [names[i] for i in range(nc.value)]
            lens = [L.cornetto_cov_lens(cov)[i] for i in range(nc.value)]
            # the name strings are malloc'd by the library: release them with libc free
            libc = C.CDLL(None)
            libc.free.argtypes = [C.c_void_p]
            arr = C.cast(names, C.POINTER(C.c_void_p))
            for i in range(nc.value):
                libc.free(arr[i])
            libc.free(C.cast(names, C.c_void_p))
            return _Resident(self, cov, L.cornetto_cov_free, lens), nm, ncl.value
        finally:
            L.cornetto_bgin_close(self.h, bg)


def khash_str_order(names):
    """bucket order of the reference's khash string map after inserting `names` (list of bytes) in order:
    (slot per name, distinct ids in bucket order).  Host only."""
    L = lib()
    n = len(names)
    arr = (C.c_char_p * max(n, 1))(*names)
    slot = np.zeros(max(n, 1), np.int32)
    order = np.zeros(max(n, 1), np.int32)
    k = L.cornetto_khash_str_order(arr, n, slot.ctypes.data, order.ctypes.data)
    if k < 0:
        raise ValueError("cornetto_khash_str_order: bad argument")
    return slot[:n], order[:k]


class PanelOpt(C.Structure):
    _fields_ = [("min_lowq_len", C.c_int32), ("extend", C.c_int32), ("edge_len", C.c_int32), ("merge_dist", C.c_int32), ("min_ctg_len", C.c_int32),
                ("extend_right", C.c_int32), ("extend_gate", C.c_int32)]


def panel_boring(ctg_len, fun, lowq, recreate=False, **kw):
    """steps 4-9 of scripts/create-cornetto.sh (recreate=True: the constants of scripts/recreate-cornetto.sh) on index-based
    intervals (host only); kw overrides PanelOpt fields — `extend` alone sets all three of extend / extend_right / extend_gate"""
    L = lib()
    opt = PanelOpt()
    (L.cornetto_panel_defaults_recreate if recreate else L.cornetto_panel_defaults)(C.byref(opt))
    if "extend" in kw:
        kw.setdefault("extend_right", kw["extend"])
        kw.setdefault("extend_gate", kw["extend"])
    for k, v in kw.items():
        setattr(opt, k, v)
    ctg_len = np.ascontiguousarray(ctg_len, dtype=np.int32)
    fun = np.ascontiguousarray(fun, dtype=IVL_DT)
    lowq = np.ascontiguousarray(lowq, dtype=IVL_DT)
    p, n = C.c_void_p(), C.c_int64()
    rc = L.cornetto_panel_boring(ctg_len.ctypes.data, len(ctg_len), fun.ctypes.data, len(fun), lowq.ctypes.data, len(lowq), C.byref(opt), C.byref(p), C.byref(n))
    if rc != 0:
        raise ValueError("cornetto_panel_boring: status %d" % rc)
    return _take(L, p, n.value, IVL_DT)
