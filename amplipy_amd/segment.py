"""Aligned-read record used on the host side of amplipy_amd.

``Segment`` carries the SAM/BAM core fields of one read and exposes the derived
properties that AmpliPy's hot path reads from ``pysam.AlignedSegment``
(reference call sites: AmpliPy.py:450-452, 460-463, 513-520, 561, 591,
700-706, 902, 910).  pysam 0.17.0 / htslib 1.13 are not vendored by the
reference and are not installable here, so the semantics below restate their
documented behaviour -- parity for these derived
properties is *unpinned* by any reference test.

The class is plain Python on purpose: it is the record type the SAM reader,
the batch packer and the golden-vector generator share.  Nothing on the GPU
path goes through it per read; batches are packed into SoA arrays
(``amplipy_amd.batch``).
"""
from __future__ import annotations

import re
from array import array

# CIGAR op codes in BAM order (AmpliPy.py:33-44 uses the same numbering).
CIGAR_OPS = "MIDNSHP=XB"
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X, OP_B = range(10)
_CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=XB])")

FLAG_PAIRED = 0x1
FLAG_UNMAPPED = 0x4
FLAG_REVERSE = 0x10


def parse_cigar(text):
    """'11M1I63M76H' -> [(0,11),(1,1),(0,63),(5,76)]; '*' -> None."""
    if text is None or text == "*":
        return None
    out = []
    pos = 0
    for m in _CIGAR_RE.finditer(text):
        if m.start() != pos:
            raise ValueError("malformed CIGAR: %r" % text)
        out.append((CIGAR_OPS.index(m.group(2)), int(m.group(1))))
        pos = m.end()
    if pos != len(text) or not out:
        raise ValueError("malformed CIGAR: %r" % text)
    return out


def format_cigar(ops):
    if ops is None:
        return "*"
    return "".join("%d%s" % (n, CIGAR_OPS[op]) for op, n in ops)


class Segment:
    """One alignment record with pysam.AlignedSegment-compatible accessors."""

    __slots__ = ("query_name", "flag", "reference_id", "reference_start", "mapping_quality",
                 "_cigar", "next_reference_id", "next_reference_start", "template_length",
                 "query_sequence", "query_qualities", "tags_raw", "rname", "rnext")

    def __init__(self, flag=0, reference_start=0, cigar=None, template_length=0,
                 query_sequence=None, query_qualities=None, query_name="*",
                 mapping_quality=255, reference_id=0, next_reference_id=-1,
                 next_reference_start=-1, tags_raw=None, rname=None, rnext=None):
        self.query_name = query_name
        self.flag = int(flag)
        self.reference_id = reference_id
        self.reference_start = int(reference_start)
        self.mapping_quality = mapping_quality
        if isinstance(cigar, str):
            cigar = parse_cigar(cigar)
        self._cigar = None if cigar is None else [(int(o), int(n)) for o, n in cigar]
        self.next_reference_id = next_reference_id
        self.next_reference_start = next_reference_start
        self.template_length = int(template_length)
        self.query_sequence = query_sequence
        if query_qualities is not None and not isinstance(query_qualities, array):
            if isinstance(query_qualities, str):
                query_qualities = array("B", (ord(c) - 33 for c in query_qualities))
            else:
                query_qualities = array("B", query_qualities)
        self.query_qualities = query_qualities
        self.tags_raw = tags_raw
        self.rname = rname
        self.rnext = rnext

    # --- flag bits -------------------------------------------------------
    @property
    def is_paired(self):
        return bool(self.flag & FLAG_PAIRED)

    @property
    def is_unmapped(self):
        return bool(self.flag & FLAG_UNMAPPED)

    @property
    def is_reverse(self):
        return bool(self.flag & FLAG_REVERSE)

    # --- CIGAR -----------------------------------------------------------
    @property
    def cigartuples(self):
        return None if self._cigar is None else list(self._cigar)

    @cigartuples.setter
    def cigartuples(self, value):
        self._cigar = None if value is None else [(int(o), int(n)) for o, n in value]

    @property
    def cigarstring(self):
        return format_cigar(self._cigar)

    # --- lengths ---------------------------------------------------------
    @property
    def query_length(self):
        return 0 if self.query_sequence is None else len(self.query_sequence)

    @property
    def reference_length(self):
        if self.is_unmapped or not self._cigar:
            return None
        rlen = sum(n for op, n in self._cigar if op in (OP_M, OP_D, OP_N, OP_EQ, OP_X))
        return rlen if rlen != 0 else 1  # htslib bam_endpos: an empty span counts as 1

    @property
    def reference_end(self):
        rlen = self.reference_length
        return None if rlen is None else self.reference_start + rlen

    @property
    def query_alignment_start(self):
        off = 0
        lq = self.query_length
        for op, n in self._cigar or ():
            if op == OP_H:
                if off != 0 and off != lq:
                    raise ValueError("Invalid clipping in CIGAR string")
            elif op == OP_S:
                off += n
            else:
                break
        return off

    @property
    def query_alignment_end(self):
        ops = self._cigar or []
        end = self.query_length
        if end == 0:
            for op, n in ops:
                if op in (OP_M, OP_I, OP_EQ, OP_X) or (op == OP_S and end == 0):
                    end += n
            return end
        lq = end
        for k in range(len(ops) - 1, 0, -1):  # index 0 is not examined
            op, n = ops[k]
            if op == OP_H:
                if end != lq:
                    raise ValueError("Invalid clipping in CIGAR string")
            elif op == OP_S:
                end -= n
            else:
                break
        return end

    @property
    def query_alignment_qualities(self):
        if self.query_length == 0:
            return None
        qs = self.query_alignment_start  # evaluated before the QUAL check, like pysam
        qe = self.query_alignment_end
        if self.query_qualities is None:
            return None
        return self.query_qualities[qs:qe]

    def get_aligned_pairs(self):
        q = 0
        r = self.reference_start
        out = []
        for op, n in self._cigar or ():
            if op in (OP_M, OP_EQ, OP_X):
                out.extend((q + k, r + k) for k in range(n))
                q += n
                r += n
            elif op in (OP_I, OP_S, OP_P):  # pysam walks P like an insertion
                out.extend((q + k, None) for k in range(n))
                q += n
            elif op in (OP_D, OP_N):
                out.extend((None, r + k) for k in range(n))
                r += n
            # H (and B) produce nothing
        return out

    def __repr__(self):
        return "Segment(%s flag=%d pos=%d cigar=%s tlen=%d)" % (
            self.query_name, self.flag, self.reference_start, self.cigarstring, self.template_length)
