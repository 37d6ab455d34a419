"""ctypes binding of libamplihip.so (include/amplihip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``amplipy_amd.build``.  There is
no Python or CPU fallback: if the shared object is missing or no GPU is visible, the calls
below raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# (development only -- AMPLIPY_DEV=1 AMPLIHIP_LIB=path: another build of the same library, for A/B timing by the tools under tools/;
#  the shipped package loads the library that lies next to this file and nothing else)
LIB_PATH = (os.environ.get("AMPLIHIP_LIB") if os.environ.get("AMPLIPY_DEV") == "1" else None) or os.path.join(_HERE, "libamplihip.so")
_LIB = None

EXPORTS = [
    "amp_version", "amp_strerror", "amp_last_error", "amp_read_status_exception", "amp_device_count",
    "amp_find_overlapping_primers", "amp_ctx_create", "amp_ctx_destroy", "amp_ctx_set_stream",
    "amp_ctx_bind_counts", "amp_set_primers", "amp_set_params", "amp_process_batch",
    "amp_process_batch_device", "amp_sync", "amp_last_kernel_ms", "amp_get_counts", "amp_add_counts",
    "amp_get_ins_events", "amp_counts_device_ptr", "amp_reduce", "amp_reset", "amp_error_reads",
    "amp_reserve_events", "amp_set_kernel_variant", "amp_set_reference", "amp_call_positions",
    "amp_event_strings", "amp_debug_counters", "amp_call_compact", "amp_debug_blocks", "amp_call_compact_view", "amp_set_timing", "amp_call_compact_begin", "amp_coordinate_helpers", "amp_drain_ins_events", "amp_set_cu_share",
    "amp_aggregate_ins_events", "amp_fast_path_active", "amp_last_kernel_variant",
]


class AmpliHipError(RuntimeError):
    def __init__(self, rc, where, detail=""):
        self.rc = rc
        super().__init__("%s: %s (%d)%s" % (where, abi.RC_NAMES.get(rc, "?"), rc, (": " + detail) if detail else ""))


def load():
    """Load libamplihip.so or raise -- never substitutes anything else."""
    global _LIB
    if _LIB is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); amplipy_amd has no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(L, name)  # AttributeError if the build is stale
        L.amp_strerror.restype = C.c_char_p
        L.amp_last_error.restype = C.c_char_p
        L.amp_last_error.argtypes = [C.c_void_p]
        L.amp_read_status_exception.restype = C.c_char_p
        L.amp_counts_device_ptr.restype = C.c_void_p
        L.amp_counts_device_ptr.argtypes = [C.c_void_p]
        L.amp_ctx_destroy.restype = None
        L.amp_ctx_destroy.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def find_overlapping_primers(ref_len, primers, offset):
    """find_overlapping_primers (AmpliPy.py:174-209) -> (min_start, max_end, max_primer_len);
    -1 stands for None."""
    L = load()
    ps = sorted((int(a), int(b)) for a, b in primers)
    st = np.array([p[0] for p in ps], np.int32); en = np.array([p[1] for p in ps], np.int32)
    mn = np.empty(ref_len, np.int32); mx = np.empty(ref_len, np.int32)
    mpl = C.c_int32(0)
    rc = L.amp_find_overlapping_primers(C.c_int32(ref_len), C.c_int32(len(ps)), C.c_void_p(abi.ptr(st)),
                                        C.c_void_p(abi.ptr(en)), C.c_int32(offset), C.c_void_p(abi.ptr(mn)),
                                        C.c_void_p(abi.ptr(mx)), C.byref(mpl))
    if rc:
        raise AmpliHipError(rc, "amp_find_overlapping_primers")
    return mn, mx, int(mpl.value)


class Engine:
    """One amp_ctx: the device-side state of a run (count table, insertion events)."""

    def __init__(self, ref_len, device=0):
        self.L = load()
        self.ref_len = int(ref_len)
        self.device = device
        h = C.c_void_p()
        rc = self.L.amp_ctx_create(C.byref(h), C.c_int(device), C.c_int32(ref_len))
        if rc:
            raise AmpliHipError(rc, "amp_ctx_create", "no usable MI355X device" if rc == -4 else "")
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.L.amp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc:
            raise AmpliHipError(rc, where, (self.L.amp_last_error(self.h) or b"").decode())

    # ---- configuration ---------------------------------------------------------------
    def set_primers(self, min_start, max_end, max_primer_len):
        mn = np.ascontiguousarray(min_start, np.int32); mx = np.ascontiguousarray(max_end, np.int32)
        assert mn.size == self.ref_len and mx.size == self.ref_len
        self._chk(self.L.amp_set_primers(self.h, C.c_void_p(abi.ptr(mn)), C.c_void_p(abi.ptr(mx)),
                                         C.c_int32(max_primer_len)), "amp_set_primers")

    def set_params(self, min_quality=20, window=4, do_trim=True, do_count=True):
        self._chk(self.L.amp_set_params(self.h, C.c_int32(min_quality), C.c_int32(window), C.c_int32(int(do_trim)),
                                        C.c_int32(int(do_count))), "amp_set_params")

    def fast_path_active(self):
        """True when runs with the current parameters take a fast kernel (windows of 1..8, min_quality <= 128)."""
        rc = self.L.amp_fast_path_active(self.h)
        if rc < 0:
            self._chk(rc, "amp_fast_path_active")
        return rc == 1

    def last_kernel_variant(self):
        """The kernel variant the last batch took (what the default, 0, resolved to)."""
        rc = self.L.amp_last_kernel_variant(self.h)
        if rc < 0:
            self._chk(rc, "amp_last_kernel_variant")
        return rc

    def set_kernel_variant(self, v):
        self._chk(self.L.amp_set_kernel_variant(self.h, C.c_int(v)), "amp_set_kernel_variant")

    def set_cu_share(self, divisor):
        """The fast kernel's grid for 1 / divisor of the CUs: for several engines in flight on different streams."""
        self._chk(self.L.amp_set_cu_share(self.h, C.c_int(divisor)), "amp_set_cu_share")

    def set_stream(self, hip_stream):
        self._chk(self.L.amp_ctx_set_stream(self.h, C.c_void_p(hip_stream)), "amp_ctx_set_stream")

    def bind_counts(self, dev_ptr):
        self._chk(self.L.amp_ctx_bind_counts(self.h, C.c_void_p(dev_ptr)), "amp_ctx_bind_counts")

    def set_timing(self, split):
        self._chk(self.L.amp_set_timing(self.h, C.c_int(1 if split else 0)), "amp_set_timing")

    def reserve_events(self, cap):
        self._chk(self.L.amp_reserve_events(self.h, C.c_int64(cap)), "amp_reserve_events")

    # ---- hot path --------------------------------------------------------------------
    def process(self, batch, read_base=0, want_trim=True):
        """amp_process_batch on a host ReadBatch; returns abi.TrimResult."""
        res = abi.TrimResult(batch)
        rd = abi.reads_struct(batch)
        out = res.struct()
        self._chk(self.L.amp_process_batch(self.h, C.byref(rd), C.c_uint64(read_base),
                                           C.byref(out) if want_trim else None), "amp_process_batch")
        return res

    def process_device(self, dev_reads, read_base=0, dev_out=None):
        """amp_process_batch_device: abi.AmpDevReads of device pointers; asynchronous."""
        self._chk(self.L.amp_process_batch_device(self.h, C.byref(dev_reads), C.c_uint64(read_base),
                                                  C.byref(dev_out) if dev_out is not None else None),
                  "amp_process_batch_device")

    def sync(self):
        self._chk(self.L.amp_sync(self.h), "amp_sync")

    def last_kernel_ms(self):
        t = C.c_float(0); s = C.c_float(0)
        self._chk(self.L.amp_last_kernel_ms(self.h, C.byref(t), C.byref(s)), "amp_last_kernel_ms")
        return float(t.value), float(s.value)

    # ---- state -----------------------------------------------------------------------
    def counts(self):
        c = np.empty((self.ref_len, abi.NSYM), np.uint32)
        self._chk(self.L.amp_get_counts(self.h, C.c_void_p(abi.ptr(c))), "amp_get_counts")
        return c

    def add_counts(self, counts):
        c = np.ascontiguousarray(counts, np.uint32)
        assert c.size == self.ref_len * abi.NSYM
        self._chk(self.L.amp_add_counts(self.h, C.c_void_p(abi.ptr(c))), "amp_add_counts")

    def counts_device_ptr(self):
        return self.L.amp_counts_device_ptr(self.h)

    def events(self):
        n = C.c_int64(0)
        self._chk(self.L.amp_get_ins_events(self.h, C.byref(n), None, C.c_int64(0)), "amp_get_ins_events")
        ev = np.zeros(int(n.value), abi.INS_EVENT_DTYPE)
        if n.value:
            self._chk(self.L.amp_get_ins_events(self.h, C.byref(n), C.c_void_p(abi.ptr(ev)), C.c_int64(ev.size)),
                      "amp_get_ins_events")
        return ev[:int(n.value)]          # the size query counts slots; a few may have been reserved and left unused

    def drain_events(self):
        """The events recorded since the last drain (or reset); the device list is empty afterwards."""
        n = C.c_int64(0)
        self._chk(self.L.amp_get_ins_events(self.h, C.byref(n), None, C.c_int64(0)), "amp_get_ins_events")
        ev = np.zeros(max(int(n.value), 1), abi.INS_EVENT_DTYPE)
        self._chk(self.L.amp_drain_ins_events(self.h, C.byref(n), C.c_void_p(abi.ptr(ev)), C.c_int64(ev.size)), "amp_drain_ins_events")
        return ev[:int(n.value)]

    def aggregate_events(self, dev_reads=None, read_base=0, drain=False):
        """amp_aggregate_ins_events: the insertion events recorded so far, sorted and run-length encoded ON THE DEVICE ->
        INS_RUN_DTYPE[n_runs], one record per (ref_pos, allele): a representative event and the number of events.
        dev_reads None = the batch of the last process() call (still staged on the device)."""
        n = C.c_int64(0)
        rdp = C.byref(dev_reads) if dev_reads is not None else None
        self._chk(self.L.amp_aggregate_ins_events(self.h, rdp, C.c_uint64(read_base), C.c_int(0), C.byref(n), None, C.c_int64(0)),
                  "amp_aggregate_ins_events")
        runs = np.zeros(max(int(n.value), 1), abi.INS_RUN_DTYPE)
        self._chk(self.L.amp_aggregate_ins_events(self.h, rdp, C.c_uint64(read_base), C.c_int(1 if drain else 0), C.byref(n),
                                                  C.c_void_p(abi.ptr(runs)), C.c_int64(runs.size)), "amp_aggregate_ins_events")
        return runs[:int(n.value)]

    def debug_blocks(self):
        out = np.zeros((4096, 4), np.uint32); nb = C.c_int(0)
        self._chk(self.L.amp_debug_blocks(self.h, C.c_void_p(abi.ptr(out)), C.c_int(4096), C.byref(nb)), "amp_debug_blocks")
        return out[:nb.value]

    def debug_counters(self):
        out = np.zeros(16, np.uint64)
        self._chk(self.L.amp_debug_counters(self.h, C.c_void_p(abi.ptr(out))), "amp_debug_counters")
        return out

    def error_reads(self):
        n = C.c_int64(0)
        self._chk(self.L.amp_error_reads(self.h, C.byref(n)), "amp_error_reads")
        return int(n.value)

    def reduce(self, comm=None, root=0):
        self._chk(self.L.amp_reduce(self.h, C.c_void_p(comm), C.c_int(root)), "amp_reduce")

    def reset(self):
        self._chk(self.L.amp_reset(self.h), "amp_reset")

    # ---- calling ---------------------------------------------------------------------
    def set_reference(self, ref_seq):
        ref = np.frombuffer(ref_seq.encode("ascii") if isinstance(ref_seq, str) else bytes(ref_seq), np.uint8)
        assert ref.size == self.ref_len
        self._chk(self.L.amp_set_reference(self.h, C.c_void_p(abi.ptr(ref))), "amp_set_reference")

    def call_positions(self, params):
        """amp_call_positions -> (structured array POS_CALL_DTYPE[ref_len], n_relevant)."""
        out = np.zeros(self.ref_len, abi.POS_CALL_DTYPE)
        nr = C.c_int64(0)
        self._chk(self.L.amp_call_positions(self.h, C.byref(params), C.c_void_p(abi.ptr(out)), C.byref(nr)),
                  "amp_call_positions")
        return out, int(nr.value)

    def coordinate_helpers(self, cigars, ref_start, ref_pos, query_pos):
        """get_pos_on_query / get_pos_on_ref / fix_cigar (AmpliPy.py:363-423) for a list of CIGARs [(op, len), ...] on the
        device -> (pos_on_query int32[n], pos_on_ref int32[n], [fixed cigar tuples], status uint8[n])."""
        n = len(cigars)
        off = np.zeros(n + 1, np.uint32)
        off[1:] = np.cumsum([len(c) for c in cigars])
        words = np.array([(int(l) << 4) | int(op) for c in cigars for op, l in c], np.uint32)
        rs, rp, qp = (np.ascontiguousarray(x, np.int32) for x in (ref_start, ref_pos, query_pos))
        oq, orf = np.zeros(n, np.int32), np.zeros(n, np.int32)
        fx, fn, st = np.zeros(max(words.size, 1), np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint8)
        self._chk(self.L.amp_coordinate_helpers(self.h, C.c_int64(n), C.c_void_p(abi.ptr(off)), C.c_void_p(abi.ptr(words) if words.size else None),
                                                C.c_void_p(abi.ptr(rs)), C.c_void_p(abi.ptr(rp)), C.c_void_p(abi.ptr(qp)),
                                                C.c_void_p(abi.ptr(oq)), C.c_void_p(abi.ptr(orf)), C.c_void_p(abi.ptr(fx)),
                                                C.c_void_p(abi.ptr(fn)), C.c_void_p(abi.ptr(st))), "amp_coordinate_helpers")
        fixed = [[(int(w & 15), int(w >> 4)) for w in fx[int(off[i]):int(off[i]) + int(fn[i])]] for i in range(n)]
        return oq, orf, fixed, st

    def call_compact_begin(self, params):
        """Enqueue the calling kernels now; the next call_compact with the same parameters only waits for them."""
        self._chk(self.L.amp_call_compact_begin(self.h, C.byref(params)), "amp_call_compact_begin")

    def call_compact(self, params):
        """amp_call_compact_view -> (consensus int8[G], VAR_REC_DTYPE[V], relevant int32[R]).  The arrays
        are views of host memory owned by the engine, valid until the next call_* on it."""
        G = self.ref_len
        if not hasattr(self, "_cv"):
            self._cv = [abi.AmpCallView(), None, None]
        st = self._cv
        v = st[0]
        rc = self.L.amp_call_compact_view(self.h, C.byref(params), C.byref(v))
        if rc:
            self._chk(rc, "amp_call_compact_view")
        key = (v.consensus, v.vars, v.relevant)
        if st[1] != key:   # (re)wrap the library's image once per allocation
            cons = np.frombuffer((C.c_int8 * G).from_address(v.consensus), np.int8)
            vars_ = np.frombuffer((C.c_uint8 * (G * abi.VAR_REC_DTYPE.itemsize)).from_address(v.vars), abi.VAR_REC_DTYPE)
            rel = np.frombuffer((C.c_int32 * G).from_address(v.relevant), np.int32)
            st[1] = key; st[2] = (cons, vars_, rel)
        cons, vars_, rel = st[2]
        return cons, vars_[:v.n_vars], rel[:v.n_relevant]

    def event_text(self, events, read_base=0, dev_reads=None):
        """(lengths int64[n], text uint8[sum]) of ``events``: the allele bytes back to back, no Python step per event.
        dev_reads None = the batch of the last process() call (still staged on the device)."""
        n = events.size
        lens = np.maximum(events["q_to"].astype(np.int64) - events["q_from"].astype(np.int64), 0)
        off = np.zeros(n + 1, np.uint64)
        np.cumsum(lens, out=off[1:])
        text = np.zeros(max(int(off[n]), 1), np.uint8)
        ev = np.ascontiguousarray(events)
        self._chk(self.L.amp_event_strings(self.h, C.byref(dev_reads) if dev_reads is not None else None, C.c_uint64(read_base), C.c_int64(n),
                                           C.c_void_p(abi.ptr(ev)), C.c_void_p(abi.ptr(off)), C.c_void_p(abi.ptr(text))),
                  "amp_event_strings")
        return lens, text[:int(off[n])]

    def event_strings_device(self, dev_reads, events, read_base=0):
        """Strings of ``events`` (INS_EVENT_DTYPE) taken from a device-resident batch."""
        n = events.size
        lens = (events["q_to"] - events["q_from"]).astype(np.uint64)
        off = np.zeros(n + 1, np.uint64)
        np.cumsum(lens, out=off[1:])
        text = np.zeros(max(int(off[n]), 1), np.uint8)
        ev = np.ascontiguousarray(events)
        self._chk(self.L.amp_event_strings(self.h, C.byref(dev_reads), C.c_uint64(read_base), C.c_int64(n),
                                           C.c_void_p(abi.ptr(ev)), C.c_void_p(abi.ptr(off)), C.c_void_p(abi.ptr(text))),
                  "amp_event_strings")
        raw = text.tobytes()
        return [raw[int(off[k]):int(off[k + 1])].decode("ascii") for k in range(n)]
