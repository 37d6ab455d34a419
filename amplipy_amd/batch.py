"""Packed structure-of-arrays read batch: the unit of work handed to the C-ABI.

Layout (include/amplihip.h documents the same contract):

  pos      int32 [n]      0-based leftmost reference coordinate (SAM POS-1)
  flag     uint16[n]      SAM FLAG (bits 0x1 paired, 0x10 reverse are used by the path)
  tlen     int32 [n]      SAM TLEN
  lseq     uint32[n]      stored SEQ length (soft clips included); 0 when SEQ is '*'
  cig_off  uint64[n+1]    op offsets into ``cig``
  cig      uint32[...]    BAM encoding ``len<<4 | op``
  seq_off  uint64[n+1]    offset of each read, in BASES, into ``qual`` (bytes) and
                          ``seq`` (nibbles); every offset is a multiple of 8 so a
                          read starts on an 8-byte qual / 4-byte seq boundary
  seq      uint8 [tot/2]  4-bit BAM base codes ("=ACMGRSVTWYHKDBN"), high nibble first
  qual     uint8 [tot]    Phred values; 0xFF in the first byte of a read = QUAL '*'

Reads the reference's driver loop skips (unmapped / no CIGAR, AmpliPy.py:902)
are dropped by the packer; ``src_index`` maps batch rows back to input order.
"""
from __future__ import annotations

import numpy as np

from .segment import Segment, format_cigar

SEQ_NT16 = "=ACMGRSVTWYHKDBN"
_NT16_LUT = np.full(256, 15, dtype=np.uint8)
for _i, _c in enumerate(SEQ_NT16):
    _NT16_LUT[ord(_c)] = _i
    _NT16_LUT[ord(_c.lower())] = _i
ALIGN = 8  # bases


def encode_seq(text):
    """ASCII bases -> 4-bit codes (htslib seq_nt16_table: unknown letters become N)."""
    return _NT16_LUT[np.frombuffer(text.encode("ascii"), dtype=np.uint8)]


def pack_nibbles(codes):
    """uint8 codes (even count) -> packed bytes, high nibble first."""
    codes = np.asarray(codes, dtype=np.uint8)
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)


def unpack_nibbles(packed, n=None):
    packed = np.asarray(packed, dtype=np.uint8)
    out = np.empty(packed.size * 2, dtype=np.uint8)
    out[0::2] = packed >> 4
    out[1::2] = packed & 15
    return out if n is None else out[:n]


class ReadBatch:
    __slots__ = ("n", "pos", "flag", "tlen", "lseq", "cig_off", "cig", "seq_off", "seq", "qual",
                 "src_index")

    def __init__(self, pos, flag, tlen, lseq, cig_off, cig, seq_off, seq, qual, src_index=None):
        self.n = int(len(pos))
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.flag = np.ascontiguousarray(flag, dtype=np.uint16)
        self.tlen = np.ascontiguousarray(tlen, dtype=np.int32)
        self.lseq = np.ascontiguousarray(lseq, dtype=np.uint32)
        self.cig_off = np.ascontiguousarray(cig_off, dtype=np.uint64)
        self.cig = np.ascontiguousarray(cig, dtype=np.uint32)
        self.seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        self.seq = np.ascontiguousarray(seq, dtype=np.uint8)
        self.qual = np.ascontiguousarray(qual, dtype=np.uint8)
        self.src_index = src_index
        self.validate()

    def validate(self):
        n = self.n
        assert self.flag.shape == (n,) and self.tlen.shape == (n,) and self.lseq.shape == (n,)
        assert self.cig_off.shape == (n + 1,) and self.seq_off.shape == (n + 1,)
        assert int(self.cig_off[n]) == self.cig.size
        tot = int(self.seq_off[n])
        assert tot % ALIGN == 0 and self.qual.size == tot and self.seq.size == tot // 2
        if n:
            assert not np.any(self.seq_off % ALIGN), "read offsets must be multiples of 8 bases"
            assert np.all(self.seq_off[1:] - self.seq_off[:-1] >= self.lseq)

    # ---- builders ---------------------------------------------------------
    @classmethod
    def from_segments(cls, segments):
        keep = [(i, s) for i, s in enumerate(segments)
                if not s.is_unmapped and s.cigartuples is not None]
        n = len(keep)
        pos = np.empty(n, np.int32); flag = np.empty(n, np.uint16); tlen = np.empty(n, np.int32)
        lseq = np.empty(n, np.uint32)
        cig_off = np.zeros(n + 1, np.uint64); seq_off = np.zeros(n + 1, np.uint64)
        cigs = []; codes = []; quals = []
        co = 0; so = 0
        for k, (_, s) in enumerate(keep):
            pos[k] = s.reference_start; flag[k] = s.flag; tlen[k] = s.template_length
            ops = s.cigartuples
            cigs.extend((ln << 4) | op for op, ln in ops)
            co += len(ops); cig_off[k + 1] = co
            L = s.query_length
            lseq[k] = L
            pad = (-L) % ALIGN
            c = np.zeros(L + pad, np.uint8)
            q = np.zeros(L + pad, np.uint8)
            if L:
                c[:L] = encode_seq(s.query_sequence)
                if s.query_qualities is None:
                    q[:L] = 0xFF
                else:
                    q[:L] = np.frombuffer(bytes(s.query_qualities), dtype=np.uint8)
            codes.append(c); quals.append(q)
            so += L + pad; seq_off[k + 1] = so
        codes = np.concatenate(codes) if codes else np.zeros(0, np.uint8)
        quals = np.concatenate(quals) if quals else np.zeros(0, np.uint8)
        return cls(pos, flag, tlen, lseq, cig_off, np.array(cigs, dtype=np.uint32), seq_off,
                   pack_nibbles(codes), quals, src_index=np.array([i for i, _ in keep], np.int64))

    @classmethod
    def from_uniform(cls, pos, flag, tlen, read_len, cig_off, cig, codes, qual):
        """All reads share ``read_len``; ``codes``/``qual`` are (n, read_len) uint8."""
        n = len(pos)
        stride = read_len + ((-read_len) % ALIGN)
        c = np.zeros((n, stride), np.uint8); q = np.zeros((n, stride), np.uint8)
        c[:, :read_len] = codes; q[:, :read_len] = qual
        seq_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(stride)
        return cls(pos, flag, tlen, np.full(n, read_len, np.uint32), cig_off, cig, seq_off,
                   pack_nibbles(c.reshape(-1)), q.reshape(-1))

    # ---- views ------------------------------------------------------------
    def segment(self, i):
        """Re-materialise row ``i`` as a Segment (tests / host tooling)."""
        a, b = int(self.cig_off[i]), int(self.cig_off[i + 1])
        ops = [(int(v) & 15, int(v) >> 4) for v in self.cig[a:b]]
        L = int(self.lseq[i]); o = int(self.seq_off[i])
        codes = unpack_nibbles(self.seq[o // 2:(o + L + 1) // 2], L)
        seq = "".join(SEQ_NT16[c] for c in codes) if L else None
        q = self.qual[o:o + L]
        quals = None if (L == 0 or q[0] == 0xFF) else q.tolist()
        return Segment(flag=int(self.flag[i]), reference_start=int(self.pos[i]), cigar=ops,
                       template_length=int(self.tlen[i]), query_sequence=seq, query_qualities=quals)

    def segments(self):
        return [self.segment(i) for i in range(self.n)]

    def slice(self, lo, hi):
        """Rows [lo, hi) as an independent batch (used to shard across ranks)."""
        c0, c1 = int(self.cig_off[lo]), int(self.cig_off[hi])
        s0, s1 = int(self.seq_off[lo]), int(self.seq_off[hi])
        return ReadBatch(self.pos[lo:hi], self.flag[lo:hi], self.tlen[lo:hi], self.lseq[lo:hi],
                         self.cig_off[lo:hi + 1] - np.uint64(c0), self.cig[c0:c1],
                         self.seq_off[lo:hi + 1] - np.uint64(s0), self.seq[s0 // 2:s1 // 2],
                         self.qual[s0:s1])

    def total_bases(self):
        return int(self.lseq.sum(dtype=np.uint64))

    def cigar_strings(self):
        return [format_cigar([(int(v) & 15, int(v) >> 4)
                              for v in self.cig[int(self.cig_off[i]):int(self.cig_off[i + 1])]])
                for i in range(self.n)]
