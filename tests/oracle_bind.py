"""ctypes bindings of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by cornetto_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")


class Hit(C.Structure):
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("strand", C.c_int32), ("pad", C.c_int32)]


class Win(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("car", C.c_int32), ("pad", C.c_int32)]


class Reg(C.Structure):
    _fields_ = [("st", C.c_int32), ("end", C.c_int32), ("depth", C.c_int32), ("mq_depth", C.c_int32)]


class FxRec(C.Structure):
    _fields_ = [("name", C.c_void_p), ("comment", C.c_void_p), ("seq", C.c_void_p), ("qual", C.c_void_p),
                ("name_l", C.c_int64), ("comment_l", C.c_int64), ("l", C.c_int64), ("qual_l", C.c_int64)]


HIT_DT = np.dtype([("start", "<i8"), ("end", "<i8"), ("strand", "<i4"), ("pad", "<i4")])
WIN_DT = np.dtype([("start", "<i4"), ("end", "<i4"), ("car", "<i4"), ("pad", "<i4")])
REG_DT = np.dtype([("st", "<i4"), ("end", "<i4"), ("depth", "<i4"), ("mq_depth", "<i4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ODIR, "liboracle.so")
        src = os.path.join(ODIR, "oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ODIR, "-s", "-B"])
        L = C.CDLL(so)
        L.orc_telofind.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(C.POINTER(Hit)), C.POINTER(C.c_int64)]
        L.orc_telofind.restype = C.c_int
        L.orc_telowin_threshold.argtypes = [C.c_double, C.c_double]
        L.orc_telowin_threshold.restype = C.c_double
        L.orc_telowin.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.POINTER(C.POINTER(Win)), C.POINTER(C.c_int64)]
        L.orc_telowin.restype = C.c_int
        L.orc_sdust.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        L.orc_sdust.restype = C.POINTER(C.c_uint64)
        L.orc_n_reg.argtypes = [C.c_int32] * 3
        L.orc_n_reg.restype = C.c_int32
        L.orc_regs_assert.argtypes = [C.c_int32] * 3
        L.orc_regs_assert.restype = C.c_int
        L.orc_get_regs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.orc_get_regs.restype = None
        L.orc_mean_depth.argtypes = [C.c_double, C.c_double]
        L.orc_mean_depth.restype = C.c_int32
        L.orc_threshold.argtypes = [C.c_float, C.c_int32]
        L.orc_threshold.restype = C.c_int32
        L.orc_is_fun.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float]
        L.orc_is_fun.restype = C.c_int
        L.orc_bigenough_keep.argtypes = [C.c_int32] * 4
        L.orc_telobreaks.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.orc_telobreaks.restype = C.c_int
        L.orc_khash_order.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_khash_order.restype = C.c_int32
        L.orc_ivl_merge.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.orc_ivl_merge.restype = C.c_int
        L.orc_panel_boring.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64] + [C.c_int32] * 7 + [C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.orc_panel_boring.restype = C.c_int
        L.orc_bigenough_keep.restype = C.c_int
        L.orc_fastx_parse.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.POINTER(FxRec)), C.POINTER(C.c_int64)]
        L.orc_fastx_parse.restype = C.c_int
        L.orc_fastx_free.argtypes = [C.POINTER(FxRec), C.c_int64]
        L.orc_fastx_free.restype = None
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_revcomp.argtypes = [C.c_char_p, C.c_char_p]
        _lib = L
    return _lib


def _buf(seq):
    """bytes / numpy uint8 -> (keepalive, pointer, length)"""
    if isinstance(seq, (bytes, bytearray)):
        a = np.frombuffer(bytes(seq), dtype=np.uint8)
    else:
        a = np.ascontiguousarray(seq, dtype=np.uint8)
    return a, a.ctypes.data, a.size


def telofind(seq, motif=b"TTAGGG"):
    """-> structured array (start,end,strand): all strand-0 runs then all strand-1 runs"""
    a, p, n = _buf(seq)
    hp = C.POINTER(Hit)()
    nh = C.c_int64()
    rc = lib().orc_telofind(p, n, motif, C.byref(hp), C.byref(nh))
    if rc != 0:
        raise ValueError("orc_telofind rc=%d" % rc)
    out = np.zeros(nh.value, dtype=HIT_DT)
    if nh.value:
        C.memmove(out.ctypes.data, hp, nh.value * C.sizeof(Hit))
    lib().orc_free(hp)
    return out


def telowin(hits, contig_len, thr_adj):
    hits = np.ascontiguousarray(hits, dtype=HIT_DT)
    wp = C.POINTER(Win)()
    nw = C.c_int64()
    lib().orc_telowin(hits.ctypes.data, hits.size, contig_len, thr_adj, C.byref(wp), C.byref(nw))
    out = np.zeros(nw.value, dtype=WIN_DT)
    if nw.value:
        C.memmove(out.ctypes.data, wp, nw.value * C.sizeof(Win))
    lib().orc_free(wp)
    return out


def telowin_threshold(thr, identity):
    return lib().orc_telowin_threshold(thr, identity)


def sdust(seq, T=20, W=64):
    """-> uint64 array of (start<<32|finish)"""
    a, p, n = _buf(seq)
    cnt = C.c_int32()
    r = lib().orc_sdust(p, n, T, W, C.byref(cnt))
    out = np.zeros(cnt.value, dtype=np.uint64)
    if cnt.value:
        C.memmove(out.ctypes.data, r, cnt.value * 8)
    lib().orc_free(r)
    return out


def n_reg(length, w, inc):
    return lib().orc_n_reg(length, w, inc)


def regs_assert(length, w, inc):
    """0, or the line of the assert of get_regs() (src/boringbits_main.c:353,368,369) that ends the reference"""
    return lib().orc_regs_assert(length, w, inc)


def get_regs(depth, mq, w, inc):
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    mq = np.ascontiguousarray(mq, dtype=np.uint16)
    n = n_reg(depth.size, w, inc)
    out = np.zeros(n, dtype=REG_DT)
    lib().orc_get_regs(depth.ctypes.data, mq.ctypes.data, depth.size, w, inc, out.ctypes.data)
    return out


def mean_depth(tot, n):
    return lib().orc_mean_depth(float(tot), float(n))


def threshold(factor, mean):
    return lib().orc_threshold(factor, mean)


def is_fun(depth, mq, lo, hi, q):
    return bool(lib().orc_is_fun(depth, mq, lo, hi, q))


def bigenough_keep(covlen, start, end, T):
    return bool(lib().orc_bigenough_keep(covlen, start, end, T))


SPAN_DT = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("end", "<i4")])
TELROW_DT = np.dtype([("ctg", "<i4"), ("start", "<i4"), ("end", "<i4"), ("matched", "<i4")])


def telobreaks(ctg_len, sd, tel):
    """sd: SPAN_DT array, tel: TELROW_DT array -> SPAN_DT array {ctg, first, last} (None: coordinates outside a contig)"""
    L = lib()
    ctg_len = np.ascontiguousarray(ctg_len, dtype=np.int32)
    sd = np.ascontiguousarray(sd, dtype=SPAN_DT)
    tel = np.ascontiguousarray(tel, dtype=TELROW_DT)
    out = C.c_void_p()
    n = C.c_int64()
    rc = L.orc_telobreaks(ctg_len.ctypes.data, len(ctg_len), sd.ctypes.data, len(sd), tel.ctypes.data, len(tel), C.byref(out), C.byref(n))
    res = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(n.value, 1) * 3,))[: n.value * 3].copy().view(SPAN_DT) if n.value else np.zeros(0, SPAN_DT)
    L.orc_free(out)
    return None if rc != 0 else res


def khash_order(names):
    """names: list of bytes -> (slot per name, ids in khash bucket order)"""
    L = lib()
    n = len(names)
    arr = (C.c_char_p * max(n, 1))(*names)
    slot = np.zeros(max(n, 1), np.int32)
    order = np.zeros(max(n, 1), np.int32)
    k = L.orc_khash_order(arr, n, slot.ctypes.data, order.ctypes.data)
    return slot[:n], order[:k]


def _spans_out(L, out, n):
    res = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(max(n.value, 1) * 3,))[: n.value * 3].copy().view(SPAN_DT) if n.value else np.zeros(0, SPAN_DT)
    L.orc_free(out)
    return res


def ivl_merge(spans, dist):
    """sort by (ctg, start) + bedtools merge -d dist"""
    L = lib()
    spans = np.ascontiguousarray(spans, dtype=SPAN_DT)
    out, n = C.c_void_p(), C.c_int64()
    L.orc_ivl_merge(spans.ctypes.data, len(spans), dist, C.byref(out), C.byref(n))
    return _spans_out(L, out, n)


def panel_boring(ctg_len, fun, lowq, min_lowq_len=8000, extend=40000, edge_len=200000, merge_dist=200000, min_ctg_len=800000,
                 extend_right=None, extend_gate=None):
    extend_right = extend if extend_right is None else extend_right
    extend_gate = extend if extend_gate is None else extend_gate
    L = lib()
    ctg_len = np.ascontiguousarray(ctg_len, dtype=np.int32)
    fun = np.ascontiguousarray(fun, dtype=SPAN_DT)
    lowq = np.ascontiguousarray(lowq, dtype=SPAN_DT)
    out, n = C.c_void_p(), C.c_int64()
    L.orc_panel_boring(ctg_len.ctypes.data, len(ctg_len), fun.ctypes.data, len(fun), lowq.ctypes.data, len(lowq), min_lowq_len, extend, edge_len,
                       merge_dist, min_ctg_len, extend_right, extend_gate, C.byref(out), C.byref(n))
    return _spans_out(L, out, n)


def fastx_parse(text):
    """kseq record framing of `text` (bytes) -> (list of (name, comment, seq, qual-or-None) bytes, final status)"""
    a, p, n = _buf(text)
    rp = C.POINTER(FxRec)()
    cnt = C.c_int64()
    rc = lib().orc_fastx_parse(p, n, C.byref(rp), C.byref(cnt))
    out = []
    for i in range(cnt.value):
        r = rp[i]
        out.append((C.string_at(r.name, r.name_l), C.string_at(r.comment, r.comment_l), C.string_at(r.seq, r.l),
                    C.string_at(r.qual, r.qual_l) if r.qual else None))
    lib().orc_fastx_free(rp, cnt.value)
    return out, rc
