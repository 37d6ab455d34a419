"""Loader for the CPU simulation of the device per-read functions (test infrastructure)."""
import ctypes as C
import os
import subprocess

from oracle import oracle

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libhostsim.so")
        src = os.path.join(_HERE, "hostsim.hip")
        hdrs = [os.path.join(_HERE, "..", "..", "amplipy_amd", "csrc", h) for h in ("amp_read.hpp", "amp_bf.hpp")]
        if not os.path.isfile(so) or os.path.getmtime(so) < max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in hdrs]):
            subprocess.check_call(["hipcc", "-O1", "-fPIC", "-shared", "--offload-arch=gfx950", "-o", so, src])
        _LIB = C.CDLL(so)
        _LIB.sim_process_range.restype = C.c_int
        _LIB.sim_free.restype = None
    return _LIB


def process(batch, ref_len, mn, mx, mpl, mq, w, do_trim=True, do_count=True, **kw):
    L = lib()
    return oracle.process(batch, ref_len, mn, mx, mpl, mq, w, do_trim=do_trim, do_count=do_count,
                          _fn=L.sim_process_range, _free=L.sim_free, **kw)


def cig2_fuzz(seed, iters):
    """(mismatches, punted, compared) of the two-segment closed forms against the generic trim code."""
    L = lib()
    L.sim_cig2_fuzz.restype = C.c_long
    np_, nc = C.c_long(0), C.c_long(0)
    bad = L.sim_cig2_fuzz(C.c_uint64(seed), C.c_long(iters), C.byref(np_), C.byref(nc))
    return int(bad), int(np_.value), int(nc.value)


def bf_fuzz(seed, iters):
    """(mismatches, punted, compared) of the branch-free closed forms (amp_bf.hpp) against the branchy ones."""
    L = lib()
    L.sim_bf_fuzz.restype = C.c_long
    np_, nc = C.c_long(0), C.c_long(0)
    bad = L.sim_bf_fuzz(C.c_uint64(seed), C.c_long(iters), C.byref(np_), C.byref(nc))
    return int(bad), int(np_.value), int(nc.value)


def regops_fuzz(seed, iters):
    """(mismatches, events, error reads) of the op-by-op indel walk (count_regular_ops) against the exact pair walk."""
    L = lib()
    L.sim_regops_fuzz.restype = C.c_long
    ne, nr = C.c_long(0), C.c_long(0)
    bad = L.sim_regops_fuzz(C.c_uint64(seed), C.c_long(iters), C.byref(ne), C.byref(nr))
    return int(bad), int(ne.value), int(nr.value)
