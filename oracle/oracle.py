"""ctypes loader for the CPU restatement (oracle/amplipy_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package amplipy_amd never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from amplipy_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.isfile(path):
            build()
        L = C.CDLL(path)
        L.orc_find_overlapping_primers.restype = C.c_int
        L.orc_pos_on_query.restype = C.c_int64
        L.orc_pos_on_ref.restype = C.c_int64
        L.orc_fix_cigar.restype = C.c_int32
        L.orc_process_range.restype = C.c_int
        L.orc_rank_alleles.restype = C.c_int32
        L.orc_free.restype = None
        _LIB = L
    return _LIB


def find_overlapping_primers(ref_len, primers, offset):
    ps = sorted((int(a), int(b)) for a, b in primers)
    st = np.array([p[0] for p in ps], np.int32); en = np.array([p[1] for p in ps], np.int32)
    mn = np.empty(ref_len, np.int32); mx = np.empty(ref_len, np.int32)
    mpl = C.c_int32(0)
    rc = lib().orc_find_overlapping_primers(C.c_int32(ref_len), C.c_int32(len(ps)), C.c_void_p(abi.ptr(st)),
                                            C.c_void_p(abi.ptr(en)), C.c_int32(offset), C.c_void_p(abi.ptr(mn)),
                                            C.c_void_p(abi.ptr(mx)), C.byref(mpl))
    assert rc == 0
    return mn, mx, int(mpl.value)


def _cig_arrays(cigar):
    ops = np.array([c[0] for c in cigar], np.int32); lens = np.array([c[1] for c in cigar], np.int64)
    return ops, lens


def pos_on_query(cigar, ref_pos, ref_start):
    ops, lens = _cig_arrays(cigar); err = C.c_int(0)
    r = lib().orc_pos_on_query(C.c_int32(len(cigar)), C.c_void_p(abi.ptr(ops)), C.c_void_p(abi.ptr(lens)),
                               C.c_int64(ref_pos), C.c_int64(ref_start), C.byref(err))
    return int(r), int(err.value)


def pos_on_ref(cigar, query_pos, ref_start):
    ops, lens = _cig_arrays(cigar); err = C.c_int(0)
    r = lib().orc_pos_on_ref(C.c_int32(len(cigar)), C.c_void_p(abi.ptr(ops)), C.c_void_p(abi.ptr(lens)),
                             C.c_int64(query_pos), C.c_int64(ref_start), C.byref(err))
    return int(r), int(err.value)


def fix_cigar(cigar):
    ops, lens = _cig_arrays(cigar)
    n = lib().orc_fix_cigar(C.c_int32(len(cigar)), C.c_void_p(abi.ptr(ops)), C.c_void_p(abi.ptr(lens)))
    return [(int(ops[i]), int(lens[i])) for i in range(n)]


class OracleResult:
    def __init__(self, trim, counts, events):
        self.trim = trim          # abi.TrimResult
        self.counts = counts      # uint32[ref_len, 6]
        self.events = events      # structured array INS_EVENT_DTYPE


def process(batch, ref_len, min_start=None, max_end=None, max_primer_len=0, min_quality=20, window=4,
            do_trim=True, do_count=True, counts=None, lo=0, hi=None, read_base=0, _fn=None, _free=None):
    """Run rows [lo, hi) of ``batch`` through the restatement (A:896-915)."""
    hi = batch.n if hi is None else hi
    res = abi.TrimResult(batch)
    if counts is None:
        counts = np.zeros((ref_len, abi.NSYM), np.uint32)
    if min_start is None:
        min_start = np.full(ref_len, -1, np.int32); max_end = np.full(ref_len, -1, np.int32)
    min_start = np.ascontiguousarray(min_start, np.int32); max_end = np.ascontiguousarray(max_end, np.int32)
    rd = abi.reads_struct(batch); out = res.struct()
    evp = C.c_void_p(); nev = C.c_int64(0)
    fn = _fn if _fn is not None else lib().orc_process_range
    rc = fn(C.c_int32(min_quality), C.c_int32(window), C.c_int32(int(do_trim)),
                                 C.c_int32(int(do_count)), C.c_int32(ref_len), C.c_void_p(abi.ptr(min_start)),
                                 C.c_void_p(abi.ptr(max_end)), C.c_int32(max_primer_len), C.byref(rd),
                                 C.c_int64(lo), C.c_int64(hi), C.c_uint64(read_base), C.byref(out),
                                 C.c_void_p(abi.ptr(counts)), C.byref(evp), C.byref(nev))
    if rc != 0:
        raise RuntimeError("oracle failed: rc=%d" % rc)
    n = int(nev.value)
    if n:
        buf = (abi.AmpInsEvent * n).from_address(evp.value)
        events = np.frombuffer(buf, dtype=abi.INS_EVENT_DTYPE, count=n).copy()
    else:
        events = np.zeros(0, abi.INS_EVENT_DTYPE)
    if evp.value:
        (_free if _free is not None else lib().orc_free)(evp)
    return OracleResult(res, counts, events)


def rank_alleles(symbols, counts):
    """sorted(((c, c/total, k) ...), reverse=True) as indices into ``symbols`` (A:771)."""
    n = len(symbols)
    arr = (C.c_char_p * n)(*[s.encode("ascii") for s in symbols])
    cnt = np.asarray(counts, np.uint32); order = np.zeros(n, np.int32); total = C.c_uint64(0)
    m = lib().orc_rank_alleles(C.c_int32(n), arr, C.c_void_p(abi.ptr(cnt)), C.c_void_p(abi.ptr(order)),
                               C.byref(total))
    return int(total.value), [int(order[i]) for i in range(m)]
