"""ctypes binding of libampbam.so (include/ampbam.h): multi-threaded BAM decode straight into a
ReadBatch, and re-encode of trimmed records.  It stands where AmpliPy.py:296-356 / :896 / :911
use pysam for BAM files; SAM text stays with bamio.py.  Host only (no GPU involved)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .batch import ReadBatch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libampbam.so")
EXPORTS = [
    "ampbam_version", "ampbam_strerror", "ampbam_open", "ampbam_close", "ampbam_last_error", "ampbam_n_records",
    "ampbam_header_text", "ampbam_n_refs", "ampbam_ref", "ampbam_decode", "ampbam_writer_open", "ampbam_write_rows",
    "ampbam_writer_close", "ampbam_writer_header_bytes", "ampbam_write_batch", "ampbam_open_range", "ampbam_open_range_at", "ampbam_part_range", "ampbam_crc32", "ampbam_inflate_raw",
]
_LIB = None


class AmpBamBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("pos", C.c_void_p), ("flag", C.c_void_p), ("tlen", C.c_void_p),
                ("lseq", C.c_void_p), ("cig_off", C.c_void_p), ("cig", C.c_void_p), ("seq_off", C.c_void_p),
                ("seq", C.c_void_p), ("qual", C.c_void_p), ("src_index", C.c_void_p),
                ("n_cig", C.c_int64), ("n_bases", C.c_int64), ("n_skipped", C.c_int64)]


def load():
    global _LIB
    if _LIB is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(L, name)
        L.ampbam_strerror.restype = C.c_char_p
        L.ampbam_last_error.restype = C.c_char_p
        L.ampbam_last_error.argtypes = [C.c_void_p]
        L.ampbam_n_records.restype = C.c_int64
        L.ampbam_n_records.argtypes = [C.c_void_p]
        L.ampbam_n_refs.argtypes = [C.c_void_p]
        L.ampbam_inflate_raw.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.ampbam_crc32.restype = C.c_uint32
        L.ampbam_crc32.argtypes = [C.c_void_p, C.c_int64]
        L.ampbam_close.restype = None
        L.ampbam_close.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


class AmpBamError(RuntimeError):
    pass


def _view(addr, dtype, count):
    """numpy view of library-owned memory (valid until the next decode / close)."""
    dt = np.dtype(dtype)
    if count == 0 or not addr:
        return np.zeros(0, dt)
    return np.frombuffer((C.c_uint8 * (count * dt.itemsize)).from_address(addr), dt)


class BamFile:
    """A BAM file inflated into host memory; records are addressed by number."""

    def __init__(self, path, threads=0, part=0, n_parts=1, first_hint=None):
        """part / n_parts: only that share of the file (ampbam_open_range: cut by compressed bytes at BGZF-block starts; records
        are numbered from the part's first one).  first_hint: the inflated offset at which the part before this one ended, when
        the caller knows it (ampbam_open_range_at: the part's first record is then not guessed)."""
        self.L = load()
        h = C.c_void_p()
        self.part, self.n_parts = int(part), int(n_parts)
        hint = 0xFFFFFFFFFFFFFFFF if first_hint is None else int(first_hint)
        rc = self.L.ampbam_open_range_at(os.fsencode(path), C.c_int(threads), C.c_int(part), C.c_int(n_parts), C.c_uint64(hint), C.byref(h))
        if rc:
            raise AmpBamError("%s: %s" % (path, self.L.ampbam_strerror(rc).decode()))
        self.h = h
        self.path = path
        self.n_records = int(self.L.ampbam_n_records(h))
        t = C.c_char_p(); n = C.c_int64()
        self.L.ampbam_header_text(h, C.byref(t), C.byref(n))
        self.header_text = C.string_at(t, n.value).decode("ascii", "replace") if n.value else ""
        self.references = []
        for i in range(int(self.L.ampbam_n_refs(h))):
            nm = C.c_char_p(); ln = C.c_int32()
            self.L.ampbam_ref(h, C.c_int32(i), C.byref(nm), C.byref(ln))
            self.references.append((nm.value.decode("ascii"), int(ln.value)))

    def part_range(self):
        """(first, end): offsets in the file's inflated stream of this part's first record and of the byte behind its last one.
        Neighbouring parts must meet: part k + 1 starts where part k ended (the check that makes the split exact)."""
        a, b = C.c_uint64(), C.c_uint64()
        self.L.ampbam_part_range(self.h, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def decode(self, first, count, copy=False):
        """Records [first, first+count) -> (ReadBatch of the rows AmpliPy.py:902 keeps, n_skipped).
        The batch's arrays are views of memory owned by this file (valid until the next decode) unless copy."""
        b = AmpBamBatch()
        rc = self.L.ampbam_decode(self.h, C.c_int64(first), C.c_int64(count), C.byref(b))
        if rc:
            raise AmpBamError("%s: %s: %s" % (self.path, self.L.ampbam_strerror(rc).decode(),
                                              (self.L.ampbam_last_error(self.h) or b"").decode()))
        n = int(b.n_reads)
        arr = (_view(b.pos, np.int32, n), _view(b.flag, np.uint16, n), _view(b.tlen, np.int32, n), _view(b.lseq, np.uint32, n),
               _view(b.cig_off, np.uint64, n + 1), _view(b.cig, np.uint32, int(b.n_cig)), _view(b.seq_off, np.uint64, n + 1),
               _view(b.seq, np.uint8, int(b.n_bases) // 2), _view(b.qual, np.uint8, int(b.n_bases)))
        src = _view(b.src_index, np.int64, n)
        if copy:
            arr = tuple(a.copy() for a in arr); src = src.copy()
        return ReadBatch(*arr, src_index=src), int(b.n_skipped)

    def close(self):
        if self.h:
            self.L.ampbam_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BamWriter:
    """Writes rows of decoded batches of ``src`` with new positions / CIGARs (AmpliPy.py:911)."""

    def __init__(self, path, header_text, src, level=-1, threads=0):
        self.L = load()
        self.src = src
        h = C.c_void_p()
        text = header_text.encode("ascii")
        rc = self.L.ampbam_writer_open(os.fsencode(path), text, C.c_int64(len(text)), src.h, C.c_int(level), C.c_int(threads), C.byref(h))
        if rc:
            raise AmpBamError("%s: %s" % (path, self.L.ampbam_strerror(rc).decode()))
        self.h = h
        self.L.ampbam_writer_header_bytes.restype = C.c_int64
        self.path = path
        self.header_bytes = int(self.L.ampbam_writer_header_bytes(h))      # the header has BGZF blocks of its own: their size in the file

    def write_rows(self, src, src_index, keep, new_pos, new_ncig, new_cig_off, new_cig):
        """Rows of a batch decoded from ``src`` (the input file, or the piece of it the batch came from; None: the file the
        writer was opened with)."""
        src = self.src if src is None else src
        src_index = np.ascontiguousarray(src_index, np.int64); keep = np.ascontiguousarray(keep, np.uint8)
        new_pos = np.ascontiguousarray(new_pos, np.int32); new_ncig = np.ascontiguousarray(new_ncig, np.uint32)
        new_cig_off = np.ascontiguousarray(new_cig_off, np.uint64); new_cig = np.ascontiguousarray(new_cig, np.uint32)
        rc = self.L.ampbam_write_rows(self.h, src.h, C.c_int64(src_index.size), C.c_void_p(src_index.ctypes.data),
                                      C.c_void_p(keep.ctypes.data), C.c_void_p(new_pos.ctypes.data),
                                      C.c_void_p(new_ncig.ctypes.data), C.c_void_p(new_cig_off.ctypes.data),
                                      C.c_void_p(new_cig.ctypes.data))
        if rc:
            raise AmpBamError("write: %s" % self.L.ampbam_strerror(rc).decode())

    def write_batch(self, b, name_base=0):
        """The rows of a packed ReadBatch as NEW records named r<name_base + row> (ampbam_write_batch: files made from synthetic
        reads; no aux fields, mate on the read's own position)."""
        arr = [np.ascontiguousarray(x, t) for x, t in ((b.pos, np.int32), (b.flag, np.uint16), (b.tlen, np.int32), (b.lseq, np.uint32),
                                                        (b.cig_off, np.uint64), (b.cig, np.uint32), (b.seq_off, np.uint64), (b.seq, np.uint8), (b.qual, np.uint8))]
        rc = self.L.ampbam_write_batch(self.h, C.c_int64(b.n), *[C.c_void_p(a.ctypes.data) for a in arr], C.c_uint64(name_base))
        if rc:
            raise AmpBamError("write_batch: %s" % self.L.ampbam_strerror(rc).decode())

    def close(self):
        if self.h:
            h, self.h = self.h, None
            rc = self.L.ampbam_writer_close(h)
            if rc:
                raise AmpBamError("close: %s" % self.L.ampbam_strerror(rc).decode())


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def stitch_bam_parts(out_path, parts):
    """The ONE trimmed BAM of a multi-rank run (AmpliPy.py:326-356, :911 write one file): ``parts`` = [(path, header_bytes)] in rank
    order, each a complete BAM written by BamWriter with the same header.  BGZF members concatenate, so the result is part 0
    without its EOF block, the later parts without their header blocks and EOF blocks, and one EOF block."""
    with open(out_path, "wb") as out:
        for k, (path, header_bytes) in enumerate(parts):
            size = os.path.getsize(path)
            with open(path, "rb") as f:
                f.seek(size - len(BGZF_EOF))
                if f.read() != BGZF_EOF:
                    raise AmpBamError("%s does not end with a BGZF EOF block" % path)
                start = 0 if k == 0 else int(header_bytes)
                f.seek(start)
                left = size - len(BGZF_EOF) - start
                while left > 0:
                    buf = f.read(min(left, 1 << 24))
                    if not buf:
                        raise AmpBamError("%s: short read" % path)
                    out.write(buf)
                    left -= len(buf)
        out.write(BGZF_EOF)
