"""ctypes mirrors of the structs and enums in include/amplihip.h (kept in the same order)."""
from __future__ import annotations

import ctypes as C

import numpy as np

NSYM = 6
SYMBOLS = "ACGTN-"
SEQ_ALIGN = 8

AMP_OK = 0
RC_NAMES = {0: "AMP_OK", -1: "AMP_EINVAL", -2: "AMP_ENOMEM", -3: "AMP_EHIP", -4: "AMP_ENODEV",
            -5: "AMP_ESTATE", -6: "AMP_EOVERFLOW", -7: "AMP_ERCCL"}

# amp_read_status -> exception class AmpliPy raises for that input
READ_STATUS_EXC = {0: None, 1: IndexError, 2: IndexError, 3: IndexError, 4: KeyError,
                   5: AttributeError, 6: TypeError, 7: ValueError, 8: IndexError, 9: TypeError}
READ_STATUS_NAMES = {0: "OK", 1: "INDEX_REF", 2: "INDEX_PAIRS", 3: "INDEX_QUERY", 4: "KEY_BASE",
                     5: "NO_SEQ", 6: "NO_QUAL", 7: "CLIP", 8: "CIGAR_OP", 9: "TYPE"}

TRIM_PRIMER_START, TRIM_PRIMER_END, TRIM_QUALITY = 1, 2, 4

INS_EVENT_DTYPE = np.dtype([("ref_pos", "<i4"), ("read", "<u4"), ("q_from", "<i4"), ("q_to", "<i4")])
# amp_ins_run: one record per (ref_pos, allele) of amp_aggregate_ins_events
INS_RUN_DTYPE = np.dtype([("ref_pos", "<i4"), ("read", "<u4"), ("q_from", "<i4"), ("q_to", "<i4"), ("count", "<u4"), ("reserved", "<u4")])


class AmpInsEvent(C.Structure):
    _fields_ = [("ref_pos", C.c_int32), ("read", C.c_uint32), ("q_from", C.c_int32), ("q_to", C.c_int32)]


class AmpReads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("pos", C.c_void_p), ("flag", C.c_void_p), ("tlen", C.c_void_p),
                ("lseq", C.c_void_p), ("cig_off", C.c_void_p), ("cig", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq", C.c_void_p), ("qual", C.c_void_p)]


class AmpTrimOut(C.Structure):
    _fields_ = [("new_pos", C.c_void_p), ("new_ncig", C.c_void_p), ("new_cig", C.c_void_p),
                ("ref_len", C.c_void_p), ("trim_flags", C.c_void_p), ("status", C.c_void_p)]


class AmpDevReads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("pos", C.c_void_p), ("flag", C.c_void_p), ("tlen", C.c_void_p),
                ("lseq", C.c_void_p), ("cig_off32", C.c_void_p), ("cig", C.c_void_p),
                ("seq_off8", C.c_void_p), ("seq", C.c_void_p), ("qual", C.c_void_p),
                ("n_cig", C.c_int64), ("n_bases_padded", C.c_int64)]


class AmpCallParams(C.Structure):
    _fields_ = [("min_depth_consensus", C.c_int32), ("min_depth_variants", C.c_int32),
                ("min_freq_consensus", C.c_double), ("min_freq_variants", C.c_double),
                ("run_consensus", C.c_int32), ("run_variants", C.c_int32),
                ("full_ranking", C.c_int32), ("reserved", C.c_int32)]


class AmpCallView(C.Structure):
    _fields_ = [("consensus", C.c_void_p), ("vars", C.c_void_p), ("relevant", C.c_void_p),
                ("n_vars", C.c_int64), ("n_relevant", C.c_int64)]


POS_CALL_DTYPE = np.dtype([("total_depth", "<u4"), ("ref_count", "<u4"), ("order", "<u4"), ("consensus_sym", "i1"),
                           ("flags", "u1"), ("alt_mask", "u1"), ("pad", "u1")])
CALL_VARIANT, CALL_GT_HAS_REF, CALL_INS_RELEVANT = 1, 2, 4
VAR_REC_DTYPE = np.dtype([("pos", "<i4"), ("total_depth", "<u4"), ("ref_count", "<u4"), ("n_alt", "u1"),
                          ("gt_has_ref", "u1"), ("alt_col", "u1", (6,)), ("alt_count", "<u4", (6,))])
assert VAR_REC_DTYPE.itemsize == 44


def ptr(a):
    """Address of a contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def reads_struct(batch):
    """AmpReads view of a ReadBatch; the batch must outlive the struct."""
    return AmpReads(batch.n, ptr(batch.pos), ptr(batch.flag), ptr(batch.tlen), ptr(batch.lseq),
                    ptr(batch.cig_off), ptr(batch.cig), ptr(batch.seq_off), ptr(batch.seq), ptr(batch.qual))


class TrimResult:
    """Host arrays receiving amp_trim_out (new CIGAR of read i at cig_off[i] + 3*i)."""

    def __init__(self, batch):
        n = batch.n
        self.batch = batch
        self.new_pos = np.zeros(n, np.int32)
        self.new_ncig = np.zeros(n, np.uint32)
        self.new_cig = np.zeros(batch.cig.size + 3 * n, np.uint32)
        self.ref_len = np.zeros(n, np.int32)
        self.trim_flags = np.zeros(n, np.uint8)
        self.status = np.zeros(n, np.uint8)

    def struct(self):
        return AmpTrimOut(ptr(self.new_pos), ptr(self.new_ncig), ptr(self.new_cig), ptr(self.ref_len),
                          ptr(self.trim_flags), ptr(self.status))

    def compact_cigars(self):
        """All reads' new CIGAR words concatenated in read order (slack between slots dropped)."""
        n = self.new_ncig.astype(np.int64)
        start = self.batch.cig_off[:-1].astype(np.int64) + 3 * np.arange(n.size, dtype=np.int64)
        tot = int(n.sum())
        if tot == 0:
            return np.zeros(0, np.uint32)
        first = np.repeat(start - (np.cumsum(n) - n), n)
        return self.new_cig[first + np.arange(tot, dtype=np.int64)]

    def cigar_ops(self, i):
        o = int(self.batch.cig_off[i]) + 3 * i
        return [(int(v) & 15, int(v) >> 4) for v in self.new_cig[o:o + int(self.new_ncig[i])]]

    def cigar_string(self, i):
        from .segment import format_cigar
        return format_cigar(self.cigar_ops(i))
